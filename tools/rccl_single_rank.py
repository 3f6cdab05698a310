#!/usr/bin/env python3
"""RCCL smoke check on a one-GPU box (development aid): the collectives bench.py issues at N > 1
(reduce of the bound accumulation tensor, all_reduce of the statistics, barrier) on a one-rank
nccl group.  Multi-rank RCCL needs one GPU per rank and is exercised by the driver's scaling runs."""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, '.')
import raytracing_amd as rt
from raytracing_amd.multi_gpu import SampleSplit, TileSplit
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
accum = torch.zeros((144, 256, 4), dtype=torch.float32, device="cuda")
color = torch.zeros((144, 256), dtype=torch.int32, device="cuda")
r = rt.Renderer(256, 144)
r.bind_buffers(accum.data_ptr(), color.data_ptr())
scene = rt.Scene.default()
r.set_scene(scene, frame_count=1)
r.dispatch(2); r.sync()
before = accum.clone()
dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
t = torch.tensor([1.0, 2.0], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
print("rccl single-rank ok", bool(torch.equal(before, accum)), t.tolist(), float(accum[..., 3].max()))
dist.destroy_process_group()
