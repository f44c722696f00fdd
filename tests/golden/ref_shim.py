"""Import the reference's own Python on CPU (build container only).

/root/reference cannot be imported as-is: ``graphs/stylegan_v2_real/op/__init__.py`` JIT-compiles CUDA at
import, ``transform_base.py`` wants torchvision / easydict / ``.cuda()``.  This module pre-registers stub
packages so that the reference's *first-party arithmetic* (networks.py, transform_base.py, pggan/model_256.py,
utils/util.py, utils/transforms.py, options/train_options.py) loads from where it lies and runs on CPU.  It
is used only by tests/golden/make_golden.py (fixture generation, build container only: the GPU box has no
/root/reference and the tests there read the committed .npz fixtures).  Nothing here is shipped product code and
no reference source text is copied: the two CUDA ops are replaced by (a) the reference's own
``upfirdn2d_native`` extracted from its file with ``ast`` at run time and (b) a leaky-relu written from the
kernel's switch table (fused_bias_act_kernel.cu:36-47).
"""
import ast
import importlib.util
import math
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF = os.environ.get('L2I_REFERENCE', '/root/reference')
SG2 = os.path.join(REF, 'graphs', 'stylegan_v2_real')


def available():
    return os.path.isfile(os.path.join(SG2, 'networks.py'))


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _native_upfirdn2d():
    """Compile the reference's own ``upfirdn2d_native`` (op/upfirdn2d.py:152-186) without importing the file."""
    path = os.path.join(SG2, 'op', 'upfirdn2d.py')
    tree = ast.parse(open(path).read(), path)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'upfirdn2d_native'][0]
    ns = {'torch': torch, 'F': F}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, 'exec'), ns)
    return ns['upfirdn2d_native']


_state = {}


def install():
    """Idempotent. Returns a namespace with the loaded reference modules."""
    if 'networks' in _state:
        return types.SimpleNamespace(**_state)
    assert available(), 'reference not present at %s' % REF
    saved = {k: sys.modules.get(k) for k in ('graphs', 'utils', 'options')}
    native = _native_upfirdn2d()

    g = _pkg('graphs', os.path.join(REF, 'graphs'))
    s = _pkg('graphs.stylegan_v2_real', SG2)
    op = _pkg('graphs.stylegan_v2_real.op', os.path.join(SG2, 'op'))
    g.stylegan_v2_real = s
    s.op = op

    def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
        b = bias.view([1, -1] + [1] * (input.dim() - 2))
        return F.leaky_relu(input + b, negative_slope) * scale

    class FusedLeakyReLU(nn.Module):          # attribute contract of op/fused_act.py:73-82
        def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
            super().__init__()
            self.bias = nn.Parameter(torch.zeros(channel))
            self.negative_slope = negative_slope
            self.scale = scale

        def forward(self, input):
            return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)

    def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
        n, c, h, w = input.shape
        out = native(input.reshape(-1, h, w, 1), kernel, up, up, down, down, pad[0], pad[1], pad[0], pad[1])
        return out.view(-1, c, out.shape[1], out.shape[2])

    op.fused_leaky_relu, op.FusedLeakyReLU, op.upfirdn2d = fused_leaky_relu, FusedLeakyReLU, upfirdn2d
    op.upfirdn2d_native = native

    networks = _load('graphs.stylegan_v2_real.networks', os.path.join(SG2, 'networks.py'))
    constants = _load('graphs.stylegan_v2_real.constants', os.path.join(SG2, 'constants.py'))
    graph_util = _load('graphs.stylegan_v2_real.graph_util', os.path.join(SG2, 'graph_util.py'))

    # --- transform_base.py: stub third-party + .cuda() -------------------------------------------------
    for name in ('torchvision', 'torchvision.utils', 'torchvision.models', 'easydict'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['torchvision'].utils = sys.modules['torchvision.utils']
    sys.modules['torchvision'].models = sys.modules['torchvision.models']
    sys.modules['easydict'].EasyDict = dict
    u = _pkg('utils', os.path.join(REF, 'utils'))
    u.image = types.ModuleType('utils.image')
    sys.modules['utils.image'] = u.image
    sg = types.ModuleType('graphs.stylegan_v2_real.stylegan2')
    sys.modules['graphs.stylegan_v2_real.stylegan2'] = sg
    s.stylegan2 = sg
    _state['orig_cuda'] = (torch.Tensor.cuda, nn.Module.cuda)
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    transform_base = _load('graphs.stylegan_v2_real.transform_base', os.path.join(SG2, 'transform_base.py'))

    # --- host-side helpers -----------------------------------------------------------------------------
    util = _load('utils.util', os.path.join(REF, 'utils', 'util.py'))
    class _Anything(types.ModuleType):         # any attribute of an absent image library resolves to None
        def __getattr__(self, item):
            if item.startswith('__'):
                raise AttributeError(item)
            return None

    for name in ('cv2', 'skimage', 'skimage.transform', 'skimage.color'):
        if name not in sys.modules:
            sys.modules[name] = _Anything(name)
    sys.modules['skimage'].transform = sys.modules['skimage.transform']
    sys.modules['skimage'].color = sys.modules['skimage.color']
    try:
        transforms = _load('utils.transforms', os.path.join(REF, 'utils', 'transforms.py'))
    except Exception as e:                         # pragma: no cover - diagnostics only
        transforms = None
        print('ref_shim: utils/transforms.py not loadable:', e)

    _pkg('graphs.pggan', os.path.join(REF, 'graphs', 'pggan'))
    model_256 = _load('graphs.pggan.model_256', os.path.join(REF, 'graphs', 'pggan', 'model_256.py'))

    _state.update(networks=networks, constants=constants, graph_util=graph_util, transform_base=transform_base,
                  util=util, transforms=transforms, model_256=model_256, op=op, saved_modules=saved)
    return types.SimpleNamespace(**_state)


def uninstall():
    """Remove the stub packages so the repo's own ``graphs`` / ``utils`` / ``options`` import normally again."""
    for k in [k for k in sys.modules if k == 'graphs' or k.startswith('graphs.') or k == 'utils'
              or k.startswith('utils.') or k == 'options' or k.startswith('options.')]:
        del sys.modules[k]
    if 'orig_cuda' in _state:
        torch.Tensor.cuda, nn.Module.cuda = _state['orig_cuda']
    _state.clear()


def load_state(module, state):
    """Copy a name->ndarray dict (latent2im_amd.synth) into a reference nn.Module, checking names & shapes."""
    sd = module.state_dict()
    assert list(sd.keys()) == list(state.keys()), (set(sd) ^ set(state))
    for k, v in state.items():
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
    module.load_state_dict({k: torch.from_numpy(v).to(sd[k].dtype) for k, v in state.items()})
    return module
