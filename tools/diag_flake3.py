import os, sys
sys.argv = [sys.argv[0], "0"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "diag_flake.py")).read().split("ref = run(")[0])
ref = run(hip_graphs=False, concurrent_streams=False)
bad = 0
for i in range(20):
    a = run()
    b = run(hip_graphs=False, concurrent_streams=False)
    if a != ref or b != ref:
        bad += 1
        print(i, "fast ok" if a == ref else "FAST differs", "plain ok" if b == ref else "PLAIN differs", flush=True)
print("mismatching rounds", bad)
