import torch
x=torch.empty(217008,128,device='cuda'); y=torch.empty(217008,256,device='cuda')
def t(fn,n=50):
    for _ in range(5): fn()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
for z in (x,y):
    ms=t(lambda: z.fill_(1.0)); print('fill %d MB: %.1f us  %.2f TB/s'%(z.numel()*4/1e6, ms*1e3, z.numel()*4/ms/1e9))
    w=torch.empty_like(z); ms=t(lambda: w.copy_(z)); print('copy %d MB: %.1f us  %.2f TB/s (r+w)'%(z.numel()*4/1e6, ms*1e3, 2*z.numel()*4/ms/1e9))
