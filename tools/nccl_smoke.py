"""World-size-1 RCCL smoke on one GPU: the exact torch.distributed calls riser_amd/dist.py makes under the nccl backend
(init with device_id and a timeout, barrier with device_ids, float64 all_reduce MAX / SUM, all_gather_object, destroy)."""
import datetime, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=60), device_id=dev)
from riser_amd import dist as rdist
rdist.barrier(dev)
print("max", rdist.reduce_scalar(1.25, "max", dev), "sum", rdist.reduce_scalar(2.5, "sum", dev))
import numpy as np
print("gather", rdist.gather_results(np.arange(3), np.ones((3, 2), np.float32), 3, dev).sum())
rdist.finalize(); print("ok")
