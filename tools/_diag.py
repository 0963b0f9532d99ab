import os, sys, time, datetime, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.argv = ["bench.py"]
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
import torch, torch.distributed as dist
import sumcheck_amd as sc
from sumcheck_amd import _lib
rank = int(os.environ.get("RANK", "0")); mode = os.environ.get("DIAG_MODE", "hip")
def cg():
    try: return " ".join(open("/sys/fs/cgroup/cpu.stat").read().split())
    except Exception as e: return str(e)
t = time.time()
if rank == 0:
    print("R0 affinity", len(os.sched_getaffinity(0)), "OMP", os.environ.get("OMP_NUM_THREADS"), "cg", cg(), flush=True)
    time.sleep(3)  # let rank 1 reach its wait
    print(subprocess.run("ps -eo pid,pcpu,nlwp,time,comm --sort=-pcpu | head -8", shell=True, capture_output=True, text=True).stdout, flush=True)
    c, _ = b.cpu_baseline([[0,1,2,3],[4,5,6],[7,8],[9]], 10, nv_full=22, budget_s=8.0)
    print("R0", mode, "cores", c["cores"], "value %.3g" % c["value"], c["sample"][:90], "one %.3g" % c["one_thread"]["value"], "wall", round(time.time()-t, 1), "cg", cg(), flush=True)
    print(subprocess.run("ps -eo pid,pcpu,nlwp,time,comm --sort=-pcpu | head -8", shell=True, capture_output=True, text=True).stdout, flush=True)
else:
    if mode == "hip":
        torch.cuda.set_device(0); _lib.check(sc.lib().sc_set_device(0))
    elif mode == "sleep":
        pass
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=30))
    ct = os.times(); print("RANK", rank, "cpu user %.1f sys %.1f wall %.1f" % (ct.user, ct.system, time.time()-t), flush=True)
    dist.barrier(); dist.destroy_process_group()
