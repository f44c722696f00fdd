"""Per-phase cycle breakdown of conv_wino_kernel (block 0, thread 0): needs a library built with -DWINO_TIMING."""
import sys, os, ctypes; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from latent2im_amd import conv, _lib
import latent2im_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.__file__), 'libl2i_hip_timing.so')
names = ['setup', 'issue', 'compute', 'commit', 'bar1', 'transform', 'bar2', 'epilogue']
for cfg in sys.argv[1:]:
    cin, cout, res, b = (int(v) for v in cfg.split(','))
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    fc = conv.FrozenConv2d(w, 1, 1, device='cuda')
    x = torch.randn(b, cin, res, res, device='cuda')
    y = torch.empty(b, cout, res, res, device='cuda')
    fc.forward(x, out=y); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    dbg = L.load().l2i_debug_wino_timing
    dbg.argtypes = [ctypes.c_void_p, ctypes.c_int]
    dbg(buf, 1)
    fc.forward(x, out=y); torch.cuda.synchronize()
    dbg(buf, 0)
    tot = sum(buf[:8])
    print(cfg, 'total cycles', tot, ' '.join('%s=%d(%.1f%%)' % (n, buf[i], 100.0 * buf[i] / tot) for i, n in enumerate(names)))
