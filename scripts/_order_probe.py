import faulthandler, sys, os, time
faulthandler.dump_traceback_later(90, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zkmi_loader import load_pkg
t0 = time.time()
z = load_pkg().Zkmi(); c = z.context(0)
print("ctx", round(time.time() - t0, 2), flush=True)
import torch
print("torch imported", round(time.time() - t0, 2), flush=True)
t = torch.arange(64, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
print("torch cuda ok", round(time.time() - t0, 2), flush=True)
d = torch.zeros(32 << 10, dtype=torch.uint8, device="cuda")
d[0] = 1; torch.cuda.synchronize()
c.ntt_dev(d.data_ptr(), 10); print("ntt queued", flush=True); c.sync(); torch.cuda.synchronize()
print("ok", int(t.sum()), round(time.time() - t0, 2), flush=True)
