"""Run the reference-faithful training step with the field and colour networks a few times (for rocprofv3 --stats)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
mode = sys.argv[1] if len(sys.argv) > 1 else "color"
frame = bench.Frame("C3", torch.device("cuda", 0), 0)
wf = {"none": False, "fields": True, "color": "color"}[mode]
for _ in range(25):
    for q in frame.params.values():
        q.grad = None
    frame.train_step(with_fields=wf)
torch.cuda.synchronize()
