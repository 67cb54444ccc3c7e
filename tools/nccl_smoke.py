import os, torch, torch.distributed as dist, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from rajepy_amd import parallel as par
os.environ["MASTER_ADDR"]="127.0.0.1"; os.environ["MASTER_PORT"]="29533"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
t=torch.arange(8,dtype=torch.float64,device="cuda").reshape(2,4)
dist.barrier()
out=[torch.empty_like(t)]; dist.all_gather(out,t); dist.all_reduce(t)
sh=par.EpochShards(np.arange(2),1)
print("nccl ok", torch.equal(out[0],t), par.gather_flux_vs_time(t, sh, 0).shape, dist.get_backend())
parts=[None]; dist.all_gather_object(parts, {"a":1}); print(parts)
dist.destroy_process_group()
