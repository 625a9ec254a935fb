import os, sys, torch, torch.distributed as dist
rank=int(os.environ['RANK']); world=int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dist.init_process_group(backend='nccl', device_id=torch.device('cuda',0))
x=torch.full((4,), float(rank), device='cuda')
out=[torch.empty_like(x) for _ in range(world)]
dist.all_gather(out, x)
torch.cuda.synchronize()
print('rank',rank,'gathered',[o[0].item() for o in out], flush=True)
if rank==0:
    r=[torch.empty(4,device='cuda') for _ in range(world)]
    dist.gather(x, r, dst=0)
    print('gather ok',[t[0].item() for t in r])
else:
    dist.gather(x, None, dst=0)
dist.destroy_process_group()
