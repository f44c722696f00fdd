"""Oracle: the "next" rows of SURVEY 8(f) — forward-only editing (apply_alpha), the attribute-preservation evaluation and
the two non-linear walk modules.  TEST INFRASTRUCTURE ONLY — see oracle/__init__.py.  Pinned against the reference's own
methods through the import shims (tests/golden/make_golden.py::gen_next -> tests/golden/next.npz).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import nets
from .step import get_alphas, get_logits, get_reg_preds, get_w, walk_linear_multi_w


def walk_mlp_multi_w(ws, alpha, P):
    """WalkMlpMultiW.forward, layers=None (transform_base.py:168-192): w_i + alpha[:, :1] * MLP(w_i);
    P = {'linear.0.weight', 'linear.0.bias', 'linear.2.*', 'linear.4.*'} (nn.Sequential numbering of :175-179)."""
    al = alpha[:, 0:1]

    def mlp(x):
        x = F.leaky_relu(F.linear(x, P['linear.0.weight'], P['linear.0.bias']), 0.2)
        x = F.leaky_relu(F.linear(x, P['linear.2.weight'], P['linear.2.bias']), 0.2)
        return F.linear(x, P['linear.4.weight'], P['linear.4.bias'])

    return [w + al * mlp(w) for w in ws]


def walk_nonlinear_w(ws, alpha, P, layers=None):
    """WalkNonLinearW.forward (transform_base.py:207-243): e = embed(alpha[:, :1] repeated 10x);
    d = Linear(LeakyReLU(Linear([e, w_i]))); layers None: w_i + d/||d||_2 (:225-229); else w_i + d for i in layers (:235-238)."""
    al = alpha[:, 0:1]
    e = F.linear(al.repeat(1, 10), P['embed.weight'], P['embed.bias'])
    out = []
    for i, w in enumerate(ws):
        if layers is not None and i not in layers:
            out.append(w)
            continue
        d = F.linear(F.leaky_relu(F.linear(torch.cat([e, w], 1), P['linear.0.weight'], P['linear.0.bias']), 0.2),
                     P['linear.2.weight'], P['linear.2.bias'])
        out.append(w + (d / torch.norm(d, dim=1, keepdim=True) if layers is None else d))
    return out


def apply_alpha(PG, PR, walk_w, z, alpha_to_graph, attr_idx, n_attr_table=40, index_=None, layers=None):
    """TransformGraph.apply_alpha, 'w' branch (transform_base.py:554-603): (edited image, alpha_org, original image).
    alpha_delta = alpha_to_graph - alpha_org; with ``index_`` the column of that attribute is overwritten by
    alpha_to_graph[:, 0] - alpha_org[:, i] (i = position of index_ in attr_idx, :580-582)."""
    with torch.no_grad():
        ws = get_w(PG, z, walk_w.shape[1])
        x0 = get_logits(PG, ws)
        a0 = get_reg_preds(PR, x0, attr_idx)
        ag = torch.as_tensor(np.asarray(alpha_to_graph), dtype=a0.dtype)
        delta = get_alphas(a0, ag).clone()
        if index_ is not None:
            if len(attr_idx) == n_attr_table:
                delta[:, index_] = ag - a0[:, index_]
            else:
                i = attr_idx.index(index_)
                delta[:, i] = ag[:, 0] - a0[:, i]
        x1 = get_logits(PG, walk_linear_multi_w(ws, delta, walk_w, layers))
    return x1, a0, x0


def clip_ims(x):
    """uint8 image of transform_base.py:713,716: uint8(clip((x + 1) / 2 * 255, 0, 255)) (truncation, not rounding)."""
    return np.uint8(np.clip(((x + 1) / 2.0) * 255, 0, 255))


def change_bucket(pred, org):
    """transform_base.py:722-738: 0 / 1 / 2 for |pred - org| <= 0.3 / 0.6 / 1, 3 = dropped."""
    out = []
    for p, o in zip(pred, org):
        d = np.abs(p - o)
        out.append(0 if d <= 0.3 else 1 if d <= 0.6 else 2 if d <= 1 else 3)
    return np.asarray(out, dtype=np.int64)


def compute_multi_attr(PG, PR, walk_w, z, alphas_to_graph, attr_idx, index_):
    """vis_multi_image_batch_alphas_compute_multi_attr (transform_base.py:675-767): all 40 regressor outputs of edited /
    original image per alpha, bucketed by the change of attribute ``index_``."""
    multi_attr, attri_org, imgs, orgs = [[], [], []], [[], [], []], [[], [], []], [[], [], []]
    for ag in alphas_to_graph:
        x1, _, x0 = apply_alpha(PG, PR, walk_w, z, ag, attr_idx, index_=index_)
        pred = nets.resnet50_forward(PR, x1).numpy()
        org = nets.resnet50_forward(PR, x0).numpy()
        b = change_bucket(pred[:, index_], org[:, index_])
        u1, u0 = clip_ims(x1.numpy()), clip_ims(x0.numpy())
        for i in range(pred.shape[0]):
            if b[i] < 3:
                multi_attr[b[i]].append(pred[i]); attri_org[b[i]].append(org[i]); imgs[b[i]].append(u1[i]); orgs[b[i]].append(u0[i])
    return multi_attr, attri_org, imgs, orgs


def attribute_preservation(multi_attrs, original_attrs, index_):
    """eval.py:221-237: per non-empty bucket the mean |edited - original| over the 39 attributes other than index_."""
    res = []
    for k in range(3):
        o, c = np.asarray(original_attrs[k]), np.asarray(multi_attrs[k])
        if o.shape[0] == 0:
            continue
        keep = [j for j in range(o.shape[1]) if j != index_]
        res.append(float(np.mean(np.abs(c[:, keep] - o[:, keep]))))
    return res
