import torch, time
dev="cuda:0"
for mb in (512, 2048, 4096):
    n=mb*1024*1024//4
    a=torch.empty(n,dtype=torch.float32,device=dev).normal_(); b=torch.empty_like(a)
    for fn,name in ((lambda: b.copy_(a),"copy_"),(lambda: torch.add(a,1.0,out=b),"add")):
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms=e0.elapsed_time(e1)/10
        print(mb,"MB",name, round(2*n*4/ms/1e6,1),"GB/s r+w")
    # read-only: sum
    for _ in range(3): a.sum()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): a.sum()
    e1.record(); torch.cuda.synchronize()
    print(mb,"MB sum (read only)", round(n*4/(e0.elapsed_time(e1)/10)/1e6,1),"GB/s")
