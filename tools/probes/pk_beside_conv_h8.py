"""Packed-fp32 instruction forms (tools/probes/pk_check_lib.hip: each checked bit-for-bit against its scalar twins inside the kernel) on stream B
while the library's conv_h8_kernel runs on stream A of the same process.  Counts mismatching results per form."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from latent2im_amd import conv
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bin', 'libpkcheck.so'))
lib.pk_check_launch.restype = ctypes.c_int
lib.pk_check_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = 'cuda'
w = torch.randn(128, 128, 3, 3) / (128 * 9) ** 0.5
hc = conv.H8Conv(w, 1, 1, device=dev)
xa = torch.randn(8, 16, 64, 64, 8, device=dev).to(torch.bfloat16); ya = torch.empty_like(xa); ba = torch.randn(128, device=dev)
seed = (0.37 + 0.001 * torch.arange(1024, device=dev)).float()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
names = ['v_pk_fma_f32 (VGPR operands)', 'v_pk_fma_f32 op_sel_hi:[1,0,1]', 'v_pk_fma_f32 op_sel:[0,1,0]', 'v_pk_mul_f32 SGPR-pair src0', 'v_pk_fma_f32 SGPR-pair src0', 'v_pk_add_f32 (VGPR operands)', 'v_pk_add_f32 op_sel_hi:[0,1]', 'v_pk_add_f32 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]', 'v_pk_mul_f32 op_sel:[1,0]']
for aggress in (True, False):
    for op, name in enumerate(names):
        bad = torch.zeros(1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        for r in range(200):
            if aggress:
                with torch.cuda.stream(sa):
                    for _ in range(4):
                        hc.forward(xa, out=ya, bias=ba, act=conv.ACT_RELU)
            with torch.cuda.stream(sb):
                rc = lib.pk_check_launch(op, bad.data_ptr(), 2000, seed.data_ptr(), 512, ctypes.c_void_p(sb.cuda_stream))
                assert rc == 0
        torch.cuda.synchronize()
        print('%-58s %s: %d mismatching results of %.1e' % (name, 'beside conv_h8' if aggress else 'alone         ', int(bad.item()), 200 * 512 * 256 * 2000 * 2.0), flush=True)
