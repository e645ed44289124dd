import os, sys
sys.argv = [sys.argv[0], "0"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "diag_flake.py")).read().split("ref = run(")[0])
print("plain        ", run(hip_graphs=False, concurrent_streams=False)[:4])
print("streams      ", run(hip_graphs=False, lookahead=False)[:4])
print("all          ", run()[:4])
