"""Which calls allocate device memory while a step is being captured into a HIP graph?  Such allocations come from a
pool private to the graph (fresh hipMalloc at capture, hipFree after the graph dies).  Prints the Python frames of every
allocation that landed in a private pool during one drop-in training run in the default (graphs + streams) mode."""
import os
import sys

sys.argv = [sys.argv[0], "0"]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import diag_fast_plain as D  # noqa: E402

D.train(True, "fast")           # warm: streams, scratch, caches
torch.cuda.memory._record_memory_history(stacks="python", max_entries=200000)
s0 = torch.cuda.memory_stats()
D.train(True, "fast")
s1 = torch.cuda.memory_stats()
snap = torch.cuda.memory._snapshot()
torch.cuda.memory._record_memory_history(enabled=None)
print("device allocations during the run:", s1["num_device_alloc"] - s0["num_device_alloc"],
      "frees:", s1["num_device_free"] - s0["num_device_free"])
seen = {}
for trace in snap["device_traces"]:
    for ev in trace:
        if ev["action"] in ("segment_alloc",):
            frames = [f"{os.path.basename(f['filename'])}:{f['line']} {f['name']}" for f in ev.get("frames", [])
                      if "cycle-gan-rna_amd" in f["filename"] or "torch/cuda/graphs" in f["filename"]][:6]
            key = (ev["size"], tuple(frames))
            seen[key] = seen.get(key, 0) + 1
for (size, frames), n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f"{n} x segment of {size >> 10} KiB allocated from:", " <- ".join(frames))
