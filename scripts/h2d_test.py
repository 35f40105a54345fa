import torch, time
dev=torch.device('cuda',0)
x=torch.empty((1000,3,227,227),dtype=torch.float32,pin_memory=True); x.uniform_()
y=torch.empty_like(x,device=dev)
torch.cuda.synchronize()
for _ in range(2):
    t=time.perf_counter(); y.copy_(x,non_blocking=True); torch.cuda.synchronize(); dt=time.perf_counter()-t
    print('pinned H2D 618 MB: %.2f ms = %.1f GB/s'%(dt*1e3, x.numel()*4/dt/1e9))
xp=torch.empty((1000,3,227,227),dtype=torch.float32); xp.uniform_()
t=time.perf_counter(); y.copy_(xp); torch.cuda.synchronize(); dt=time.perf_counter()-t
print('pageable H2D: %.2f ms'%(dt*1e3))
