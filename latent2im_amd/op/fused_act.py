"""fused bias + leaky-ReLU with the reference's autograd contract (op/fused_act.py:19-86) on l2i_fused_bias_act_f32."""
import torch
from torch import nn
from torch.autograd import Function

from .. import kernels


class _FusedLeakyReLUBackward(Function):
    # grad_input = fused_bias_act(grad_output, empty, out, act=3, grad=1); grad_bias = sum over all but dim 1
    @staticmethod
    def forward(ctx, grad_output, out, negative_slope, scale):
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        grad_input = kernels.fused_bias_act(grad_output.contiguous(), None, out, 3, 1, negative_slope, scale)
        dims = [0] + list(range(2, grad_input.dim()))
        return grad_input, grad_input.sum(dims).detach()

    @staticmethod
    def backward(ctx, gradgrad_input, gradgrad_bias):
        out, = ctx.saved_tensors
        gg = kernels.fused_bias_act(gradgrad_input.contiguous(), gradgrad_bias, out, 3, 1, ctx.negative_slope, ctx.scale)
        return gg, None, None, None


class _FusedLeakyReLU(Function):
    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        out = kernels.fused_bias_act(input, bias, None, 3, 0, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        return out

    @staticmethod
    def backward(ctx, grad_output):
        out, = ctx.saved_tensors
        grad_input, grad_bias = _FusedLeakyReLUBackward.apply(grad_output, out, ctx.negative_slope, ctx.scale)
        return grad_input, grad_bias, None, None


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    return _FusedLeakyReLU.apply(input, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
