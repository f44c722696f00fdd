"""upfirdn2d with the reference's autograd contract (op/upfirdn2d.py:18-149) on l2i_upfirdn2d_f32: the backward is the
same op with the flipped FIR, up and down exchanged and the gradient pads of op/upfirdn2d.py:110-115."""
import torch
from torch.autograd import Function

from .. import kernels


class _UpFirDn2dBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, kernel, grad_kernel, up, down, pad, g_pad, in_size, out_size):
        ctx.save_for_backward(kernel)
        ctx.up, ctx.down, ctx.pad, ctx.in_size, ctx.out_size = up, down, pad, in_size, out_size
        gi = kernels.upfirdn2d(grad_output.contiguous(), grad_kernel, up=down, down=up, pad=g_pad)
        return gi.view(in_size[0], in_size[1], in_size[2], in_size[3])

    @staticmethod
    def backward(ctx, gradgrad_input):
        kernel, = ctx.saved_tensors
        gg = kernels.upfirdn2d(gradgrad_input.contiguous(), kernel, up=ctx.up, down=ctx.down, pad=ctx.pad)
        return gg, None, None, None, None, None, None, None, None


class _UpFirDn2d(Function):
    @staticmethod
    def forward(ctx, input, kernel, up, down, pad):
        up_x, up_y = up
        down_x, down_y = down
        pad_x0, pad_x1, pad_y0, pad_y1 = pad
        kernel_h, kernel_w = kernel.shape
        _, _, in_h, in_w = input.shape
        ctx.in_size = input.shape
        out = kernels.upfirdn2d(input, kernel, up=up, down=down, pad=pad)
        out_h, out_w = out.shape[2], out.shape[3]
        ctx.save_for_backward(kernel, torch.flip(kernel, [0, 1]).contiguous())
        ctx.out_size = (out_h, out_w)
        ctx.up, ctx.down, ctx.pad = up, down, pad
        ctx.g_pad = (kernel_w - pad_x0 - 1, in_w * up_x - out_w * down_x + pad_x0 - up_x + 1,
                     kernel_h - pad_y0 - 1, in_h * up_y - out_h * down_y + pad_y0 - up_y + 1)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        kernel, grad_kernel = ctx.saved_tensors
        gi = _UpFirDn2dBackward.apply(grad_output, kernel, grad_kernel, ctx.up, ctx.down, ctx.pad, ctx.g_pad,
                                      ctx.in_size, ctx.out_size)
        return gi, None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    return _UpFirDn2d.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
