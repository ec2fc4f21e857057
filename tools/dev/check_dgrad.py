import sys, torch
sys.path.insert(0, '/root/repo/efficient-attention_amd')
from efficient_attention import _ops, _native as nv
torch.manual_seed(0)
dev='cuda'
for rows in (100352, 25088, 1000, 33):
  for cd in (torch.bfloat16, torch.float16):
    for xd in (torch.float32, cd):
      for wf32 in (True, False):
        dy = (torch.randn(rows, 576, device=dev) * 0.5).to(cd)
        w = torch.randn(576, 192, device=dev) * 0.05
        w16 = None if wf32 else w.to(cd)
        _ops.DGRAD_RS_MIN_ROWS = 1
        dx = _ops.qkv_dgrad(dy, w, w16, xd)
        ref = (dy.double() @ w.to(cd).double())
        err = (dx.double() - ref).abs().max().item() / ref.abs().max().item()
        lib = _ops._mm_out(dy, w.to(cd), xd)
        same = torch.equal(lib, dx)
        print(rows, cd, xd, wf32, 'relerr %.2e' % err, 'bit-equal-to-library', same, 'maxdiff-lib %.2e' % (lib.double()-dx.double()).abs().max().item())
        assert err < (2e-3 if xd == torch.float32 else 1e-2)
# timing
import time
dy = (torch.randn(100352, 576, device=dev) * 0.5).to(torch.bfloat16)
w = torch.randn(576, 192, device=dev) * 0.05
w16 = w.to(torch.bfloat16)
for name, fn in (('own_w16', lambda: _ops.qkv_dgrad(dy, w, w16, torch.float32)), ('own_w32', lambda: _ops.qkv_dgrad(dy, w, None, torch.float32)), ('lib', lambda: _ops._mm_out(dy, w16, torch.float32))):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, 'cfg3 us', e0.elapsed_time(e1) / 50 * 1000)
dy = (torch.randn(25088, 576, device=dev) * 0.5).to(torch.bfloat16)
for name, fn in (('own_w16', lambda: _ops.qkv_dgrad(dy, w, w16, torch.float32)), ('own_w32', lambda: _ops.qkv_dgrad(dy, w, None, torch.float32)), ('lib', lambda: _ops._mm_out(dy, w16, torch.float32))):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, 'cfg2 us', e0.elapsed_time(e1) / 50 * 1000)
