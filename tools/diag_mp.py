import os, sys, time
mode = sys.argv[1]
rank = int(os.environ.get("RANK", "0"))
print(rank, "env", {k: v for k, v in os.environ.items() if "VISIBLE" in k or "HSA" in k or "ROCR" in k}, flush=True)
if mode == "lib_first":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from treegp_amd import _lib
    print(rank, "tgp devices", _lib.load_library().tgp_device_count(), flush=True)
    ctx = _lib.get_ctx(device=0)
    print(rank, "ctx ok", flush=True)
import torch
print(rank, "torch count", torch.cuda.device_count(), "avail", torch.cuda.is_available(), flush=True)
torch.cuda.set_device(0)
x = torch.ones(4, device="cuda")
print(rank, "torch ok", float(x.sum()), flush=True)
time.sleep(2)
