import torch, time
def tm(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
T=100352
for (I,O) in ((192,576),(192,192)):
    x=torch.randn(T,I,device="cuda",dtype=torch.bfloat16); w=torch.randn(O,I,device="cuda",dtype=torch.bfloat16)*0.05
    b=torch.randn(O,device="cuda",dtype=torch.bfloat16); dy=torch.randn(T,O,device="cuda",dtype=torch.bfloat16)
    wt=w.t().contiguous()
    print("shape in=%d out=%d"%(I,O))
    print("  fwd F.linear            %.1f us" % tm(lambda: torch.nn.functional.linear(x,w,b)))
    print("  fwd addmm(wt)           %.1f us" % tm(lambda: torch.addmm(b, x, wt)))
    print("  dgrad dy@w              %.1f us" % tm(lambda: dy@w))
    print("  dgrad linear(dy, wt)    %.1f us" % tm(lambda: torch.nn.functional.linear(dy, wt)))
    print("  wgrad dy.t()@x          %.1f us" % tm(lambda: dy.t()@x))
    for S in (7,14,28,49,98,196,392):
        if T % S: continue
        f=lambda: torch.bmm(dy.view(S,T//S,O).transpose(1,2), x.view(S,T//S,I))
        print("  wgrad bmm S=%3d         %.1f us" % (S, tm(f)))
    xt = x.t().contiguous(); dyt = dy.t().contiguous()
    print("  wgrad (pre-transposed dyt@x) %.1f us" % tm(lambda: dyt@x))
    S=49
    print("  db sum bf16->f32        %.1f us" % tm(lambda: dy.sum(0,dtype=torch.float32)))
    ones=torch.ones(1,T,device="cuda",dtype=torch.bfloat16)
    print("  db ones@dy              %.1f us" % tm(lambda: ones@dy))
    print("  cast x fp32->bf16       %.1f us" % tm(lambda: x.float().to(torch.bfloat16)))
