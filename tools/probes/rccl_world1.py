"""RCCL accepts what the sharded optimizer asks of it (one rank, one GPU: the only RCCL run a one-GPU box allows): in-place
reduce_scatter_tensor (output = this rank's slot of the input) with ReduceOp.AVG in f32 and bf16, in-place all_gather_into_tensor on a
side stream with an event, all_reduce / broadcast / barrier.  python tools/probes/rccl_world1.py"""
import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", world_size=1, rank=0)
n = 1 << 20
buf = torch.randn(n, device="cuda")
ref = buf.clone()
# in-place reduce-scatter: output = this rank's slot of the input
w = dist.reduce_scatter_tensor(buf[0:n], buf, op=dist.ReduceOp.AVG, async_op=True); w.wait()
torch.cuda.synchronize(); assert torch.equal(buf, ref)
bb = buf.bfloat16(); w = dist.reduce_scatter_tensor(bb[0:n], bb, op=dist.ReduceOp.AVG, async_op=True); w.wait()
# in-place all-gather: input = this rank's slot of the output
dist.all_gather_into_tensor(buf, buf[0:n]); torch.cuda.synchronize(); assert torch.equal(buf, ref)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    dist.all_gather_into_tensor(bb, bb[0:n]); ev = torch.cuda.Event(); ev.record()
torch.cuda.current_stream().wait_event(ev)
dist.all_reduce(buf[:16]); dist.broadcast(buf, src=0); dist.barrier()
print("rccl world-1 collectives ok", torch.__version__)
dist.destroy_process_group()
