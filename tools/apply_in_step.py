"""Put the winners of tools/tune_in_step.py into csrc/gemm_tuned.h: an entry of the same shape is replaced, a shape that
had no entry (it used the cost model) gets one.

    python tools/apply_in_step.py cycle-gan-rna_amd/csrc/gemm_tuned.h tuned_f32 tools/gemm_in_step_f32.csv
"""
import re
import sys

path, table, rows = sys.argv[1:4]
s = open(path).read()
start = s.index("static const TunedGemm %s[] = {" % table)
end = s.index("};", start)
body = s[start:end]
for line in open(rows):
    if line.startswith("#") or not line.strip():
        continue
    f = line.strip().split(",")
    key = "{%s, %s, %s, %s, %s, %s, " % tuple(f[:6])
    new = "    %s%s, %s},   // %s us in the step" % (key, f[6], f[7], f[8])
    pat = re.compile(r"    " + re.escape(key) + r"\d+, \d+\},\s+// [0-9.]+ us[^\n]*")
    if pat.search(body):
        body = pat.sub(new, body)
    else:
        body = body + new + "\n"
open(path, "w").write(s[:start] + body + s[end:])
