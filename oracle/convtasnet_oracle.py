"""CPU oracle for the Conv-TasNet hot path.  TEST INFRASTRUCTURE ONLY.

A functional restatement, on torch CPU tensors, of what the reference computes
for the Conv-TasNet forward, the utterance-level PIT SI-SDR loss and one
training step.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this file; the product package never does.

Parity pin: the reference has no golden vectors of its own, so this oracle is
pinned by outputs of the reference itself run in the build container
(tools/gen_golden.py -> tests/golden/*.npz, checked by
tests/test_oracle_golden.py).

Each function cites the reference file:line it restates (paths relative to the
reference root).  Parameters are addressed by the reference's state_dict keys.
Every function is dtype-generic: feed float64 tensors for a high-precision run.
"""
import itertools
import math

import torch
import torch.nn.functional as F

EPS = 1e-12


# --------------------------------------------------------------------------
# norms: src/modules/norm.py
# --------------------------------------------------------------------------
def gln(x, weight, bias, eps=EPS):
    """GlobalLayerNorm = nn.GroupNorm(1, C, eps): src/modules/norm.py:18,27."""
    return F.group_norm(x, 1, weight, bias, eps)


def gln_manual(x, weight, bias, eps=EPS):
    """Same quantity spelled out (biased variance, eps inside the sqrt)."""
    dims = tuple(range(1, x.dim()))
    mean = x.mean(dim=dims, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=dims, keepdim=True)
    shape = (1, -1) + (1,) * (x.dim() - 2)
    return (x - mean) / torch.sqrt(var + eps) * weight.view(shape) + bias.view(shape)


def cln(x, gamma, beta, eps=EPS):
    """CumulativeLayerNorm1d: src/modules/norm.py:58-95 (eps outside the sqrt, :90)."""
    shape = x.shape
    if x.dim() == 4:  # (B, C, S, chunk) -> (B, C, S*chunk), norm.py:71-74
        x = x.reshape(shape[0], shape[1], -1)
    B, C, T = x.shape
    step_sum = x.sum(dim=1)                      # :78
    step_sq = (x ** 2).sum(dim=1)                # :79
    cum_sum = torch.cumsum(step_sum, dim=1)      # :80
    cum_sq = torch.cumsum(step_sq, dim=1)        # :81
    cum_num = torch.arange(C, C * (T + 1), C, dtype=x.dtype, device=x.device)  # :83
    cum_mean = cum_sum / cum_num                 # :84
    cum_var = cum_sq / cum_num - cum_mean ** 2   # :85-86
    out = (x - cum_mean.unsqueeze(1)) / (torch.sqrt(cum_var.unsqueeze(1)) + eps) * gamma + beta  # :90
    return out.reshape(shape)


def layer_norm(x, sd, prefix, causal, eps=EPS):
    """choose_layer_norm dispatch: src/utils/tasnet.py:14-32."""
    if causal:
        return cln(x, sd[prefix + "gamma"], sd[prefix + "beta"], eps)
    return gln(x, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], eps)


# --------------------------------------------------------------------------
# encoder / decoder: src/models/filterbank.py
# --------------------------------------------------------------------------
def encoder(x, weight, stride, nonlinear=None):
    """Encoder.forward: src/models/filterbank.py:222-230 (Conv1d bias=False + optional ReLU)."""
    w = F.conv1d(x, weight, None, stride=stride)
    if nonlinear == "relu":
        w = F.relu(w)
    elif nonlinear is not None:
        raise NotImplementedError(nonlinear)
    return w


def decoder(w_hat, weight, stride):
    """Decoder.forward: src/models/filterbank.py:245-247 (ConvTranspose1d bias=False)."""
    return F.conv_transpose1d(w_hat, weight, None, stride=stride)


# --------------------------------------------------------------------------
# TCN: src/models/tdcn.py
# --------------------------------------------------------------------------
def residual_block(x, sd, prefix, dilation, causal, dual_head, kernel_size=3, eps=EPS, stride=1, separable=True):
    """ResidualBlock1d.forward (prelu, norm): src/models/tdcn.py:107-147, with the dual-head DepthwiseSeparableConv1d
    of tdcn.py:177-196 inlined; separable=False takes the dense dilated heads of :102-105 and :136-142."""
    T = x.shape[-1]
    residual = x
    y = F.conv1d(x, sd[prefix + "bottleneck_conv1d.weight"], sd[prefix + "bottleneck_conv1d.bias"])  # :116
    y = F.prelu(y, sd[prefix + "nonlinear1d.weight"])                                               # :119
    y = layer_norm(y, sd, prefix + "norm1d.", causal, eps)                                          # :121
    padding = (T - 1) * stride - T + (kernel_size - 1) * dilation + 1                               # :123
    if causal:
        pad_l, pad_r = padding, 0                                                                   # :125-127
    else:
        pad_l = padding // 2
        pad_r = padding - pad_l                                                                     # :129-130
    y = F.pad(y, (pad_l, pad_r))                                                                    # :132
    if not separable:
        out = None
        if dual_head:
            out = F.conv1d(y, sd[prefix + "output_conv1d.weight"], sd[prefix + "output_conv1d.bias"],
                           dilation=dilation) + residual                                            # :137, :145
        skip = F.conv1d(y, sd[prefix + "skip_conv1d.weight"], sd[prefix + "skip_conv1d.bias"], dilation=dilation)  # :142
        assert skip.shape[-1] == T
        return out, skip
    sp = prefix + "separable_conv1d."
    H = y.shape[1]
    y = F.conv1d(y, sd[sp + "depthwise_conv1d.weight"], sd[sp + "depthwise_conv1d.bias"],
                 stride=stride, dilation=dilation, groups=H)                                        # :181
    y = F.prelu(y, sd[sp + "nonlinear1d.weight"])                                                   # :184
    y = layer_norm(y, sd, sp + "norm1d.", causal, eps)                                              # :187
    if dual_head:
        out = F.conv1d(y, sd[sp + "output_pointwise_conv1d.weight"], sd[sp + "output_pointwise_conv1d.bias"])  # :190
        out = out + residual                                                                        # :145
    else:
        out = None
    skip = F.conv1d(y, sd[sp + "skip_pointwise_conv1d.weight"], sd[sp + "skip_pointwise_conv1d.bias"])         # :194
    assert skip.shape[-1] == T
    return out, skip


def tdcn(x, sd, prefix, num_blocks, num_layers, causal, kernel_size=3, eps=EPS, dilated=True, separable=True):
    """TimeDilatedConvNet.forward + TimeDilatedConvBlock1d.forward: src/models/tdcn.py:29-41, :65-75.
    Dilation 2**layer at stride 1 (:52-54), or dilation 1 at stride 2 when not dilated (:55-57); the very last layer has
    no residual head (:22-23, :58-59)."""
    skip_total = 0
    for r in range(num_blocks):
        block_skip = 0
        for l in range(num_layers):
            dual = not (r == num_blocks - 1 and l == num_layers - 1)
            x, skip = residual_block(x, sd, "{}net.{}.net.{}.".format(prefix, r, l), 2 ** l if dilated else 1, causal,
                                     dual, kernel_size, eps, stride=1 if dilated else 2, separable=separable)
            block_skip = block_skip + skip
        skip_total = skip_total + block_skip
    return skip_total


# --------------------------------------------------------------------------
# separator + model: src/models/conv_tasnet.py
# --------------------------------------------------------------------------
def separator(w, sd, cfg, prefix="separator."):
    """Separator.forward: src/models/conv_tasnet.py:359-378."""
    B, N, Tp = w.shape
    causal = cfg["causal"]
    eps = cfg.get("eps", EPS)
    x = layer_norm(w, sd, prefix + "norm1d.", causal, eps)                                          # :370
    x = F.conv1d(x, sd[prefix + "bottleneck_conv1d.weight"], sd[prefix + "bottleneck_conv1d.bias"])  # :371
    x = tdcn(x, sd, prefix + "tdcn.", cfg["sep_num_blocks"], cfg["sep_num_layers"], causal,
             cfg.get("sep_kernel_size", 3), eps, dilated=cfg.get("dilated", True),
             separable=cfg.get("separable", True))                                                  # :372
    x = F.prelu(x, sd[prefix + "prelu.weight"])                                                     # :373
    x = F.conv1d(x, sd[prefix + "mask_conv1d.weight"], sd[prefix + "mask_conv1d.bias"])             # :374
    if cfg.get("mask_nonlinear", "sigmoid") == "sigmoid":
        x = torch.sigmoid(x)                                                                        # :375
    elif cfg["mask_nonlinear"] == "softmax":
        x = torch.softmax(x, dim=1)   # over all n_sources*N channels jointly, conv_tasnet.py:346,375
    else:
        raise ValueError(cfg["mask_nonlinear"])
    return x.view(B, cfg["n_sources"], N, Tp)                                                       # :376


def conv_tasnet_forward(x, sd, cfg, return_latent=False):
    """ConvTasNet.extract_latent: src/models/conv_tasnet.py:121-171 (3-D input, trainable enc/dec)."""
    B, C_in, T = x.shape
    assert C_in == 1
    L = cfg["kernel_size"]
    S = cfg.get("stride") or L // 2                                                                 # :69-70
    n_src, N = cfg["n_sources"], cfg["n_basis"]
    padding = (S - (T - L) % S) % S                                                                 # :145
    pad_l = padding // 2
    pad_r = padding - pad_l                                                                         # :146-147
    xp = F.pad(x, (pad_l, pad_r))                                                                   # :149
    w = encoder(xp, sd["encoder.conv1d.weight"], S, cfg.get("enc_nonlinear"))                       # :150
    mask = separator(w, sd, cfg)                                                                    # :158
    w_hat = w.unsqueeze(1) * mask                                                                   # :159-160
    x_hat = decoder(w_hat.reshape(B * n_src, N, -1), sd["decoder.conv_transpose1d.weight"], S)      # :163-164
    x_hat = x_hat.view(B, n_src, -1)                                                                # :166
    out = F.pad(x_hat, (-pad_l, -pad_r))                                                            # :169
    if return_latent:
        return out, w_hat
    return out


# --------------------------------------------------------------------------
# SI-SDR + PIT: src/criterion/sdr.py, src/criterion/pit.py
# --------------------------------------------------------------------------
def sisdr(est, tgt, eps=EPS):
    """sisdr: src/criterion/sdr.py:135-137."""
    alpha = (est * tgt).sum(dim=-1, keepdim=True) / ((tgt ** 2).sum(dim=-1, keepdim=True) + eps)
    ratio = (((alpha * tgt) ** 2).sum(dim=-1) + eps) / (((alpha * tgt - est) ** 2).sum(dim=-1) + eps)
    return 10 * torch.log10(ratio)


def neg_sisdr(est, tgt, batch_mean=True, reduction="mean", eps=EPS):
    """NegSISDR.forward: src/criterion/sdr.py:198-227."""
    loss = -sisdr(est, tgt, eps)
    if reduction and est.dim() == 3:
        loss = loss.mean(dim=1) if reduction == "mean" else loss.sum(dim=1)
    if batch_mean:
        loss = loss.mean(dim=0)
    return loss


def pit(criterion, est, tgt, batch_mean=True, maximize=False):
    """pit: src/criterion/pit.py:9-44.  The TARGET is permuted (:30); ties keep the first
    permutation in itertools order (:37-39)."""
    n_src = est.shape[1]
    patterns = torch.tensor(list(itertools.permutations(range(n_src))), dtype=torch.long)
    losses = torch.stack([criterion(est, tgt[:, p], batch_mean=False) for p in patterns], dim=1)
    if maximize:
        loss, idx = torch.max(losses, dim=1)
    else:
        loss, idx = torch.min(losses, dim=1)
    if batch_mean:
        loss = loss.mean(dim=0)
    return loss, patterns[idx]


def pit_neg_sisdr(est, tgt, batch_mean=True):
    """PIT1d(NegSISDR()) as wired in egs/wsj0-mix/conv-tasnet/local/train.py:114-118."""
    return pit(neg_sisdr, est, tgt, batch_mean=batch_mean, maximize=False)


# --------------------------------------------------------------------------
# training step: egs/wsj0-mix/common/src/driver.py:141-157
# --------------------------------------------------------------------------
def clip_grad_norm_(grads, max_norm):
    """nn.utils.clip_grad_norm_ (driver.py:152-153): scale by max_norm/(norm+1e-6) if < 1."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).to(grads[0].dtype)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


class AdamState:
    """torch.optim.Adam(lr=1e-3) defaults (local/train.py:106): betas (0.9, 0.999), eps 1e-8."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.lr, self.betas, self.eps = lr, betas, eps
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = 0

    def step(self, params, grads):
        self.t += 1
        b1, b2 = self.betas
        bc1 = 1 - b1 ** self.t
        bc2 = 1 - b2 ** self.t
        for k, p in params.items():
            g = grads[k]
            self.m[k].mul_(b1).add_(g, alpha=1 - b1)
            self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(self.m[k], denom, value=-self.lr / bc1)


def loss_and_grads(mixture, sources, sd, cfg):
    """forward -> PIT loss -> backward (driver.py:146-150).  Returns loss, perm, output, grads."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    out = conv_tasnet_forward(mixture, params, cfg)
    loss, perm = pit_neg_sisdr(out, sources)
    grads = torch.autograd.grad(loss, list(params.values()))
    return loss.detach(), perm, out.detach(), dict(zip(params.keys(), grads))


def train_step(mixture, sources, sd, cfg, opt, max_norm=5.0):
    """One optimizer step exactly as driver.py:146-155; sd is updated in place."""
    loss, perm, _, grads = loss_and_grads(mixture, sources, sd, cfg)
    glist = [grads[k] for k in sd]
    if max_norm:
        clip_grad_norm_(glist, max_norm)
    with torch.no_grad():
        opt.step(sd, grads)
    return loss, perm


# --------------------------------------------------------------------------
# parameter shapes (what ConvTasNet.__init__ registers, in state_dict order)
# --------------------------------------------------------------------------
def parameter_shapes(cfg):
    """Keys and shapes of ConvTasNet(...).state_dict() for trainable enc/dec, separable,
    dilated, prelu, norm (src/models/conv_tasnet.py:57-114, src/models/tdcn.py)."""
    N, L = cfg["n_basis"], cfg["kernel_size"]
    Bc, H, Sc = cfg["sep_bottleneck_channels"], cfg["sep_hidden_channels"], cfg["sep_skip_channels"]
    P = cfg.get("sep_kernel_size", 3)
    R, X = cfg["sep_num_blocks"], cfg["sep_num_layers"]
    causal, n_src = cfg["causal"], cfg["n_sources"]
    shapes = {}

    def norm(prefix, C):
        if causal:
            shapes[prefix + "gamma"] = (1, C, 1)
            shapes[prefix + "beta"] = (1, C, 1)
        else:
            shapes[prefix + "norm.weight"] = (C,)
            shapes[prefix + "norm.bias"] = (C,)

    shapes["encoder.conv1d.weight"] = (N, 1, L)
    norm("separator.norm1d.", N)
    shapes["separator.bottleneck_conv1d.weight"] = (Bc, N, 1)
    shapes["separator.bottleneck_conv1d.bias"] = (Bc,)
    for r in range(R):
        for l in range(X):
            p = "separator.tdcn.net.{}.net.{}.".format(r, l)
            shapes[p + "bottleneck_conv1d.weight"] = (H, Bc, 1)
            shapes[p + "bottleneck_conv1d.bias"] = (H,)
            shapes[p + "nonlinear1d.weight"] = (1,)
            norm(p + "norm1d.", H)
            s = p + "separable_conv1d."
            shapes[s + "depthwise_conv1d.weight"] = (H, 1, P)
            shapes[s + "depthwise_conv1d.bias"] = (H,)
            shapes[s + "nonlinear1d.weight"] = (1,)
            norm(s + "norm1d.", H)
            if not (r == R - 1 and l == X - 1):
                shapes[s + "output_pointwise_conv1d.weight"] = (Bc, H, 1)
                shapes[s + "output_pointwise_conv1d.bias"] = (Bc,)
            shapes[s + "skip_pointwise_conv1d.weight"] = (Sc, H, 1)
            shapes[s + "skip_pointwise_conv1d.bias"] = (Sc,)
    shapes["separator.prelu.weight"] = (1,)
    shapes["separator.mask_conv1d.weight"] = (n_src * N, Sc, 1)
    shapes["separator.mask_conv1d.bias"] = (n_src * N,)
    shapes["decoder.conv_transpose1d.weight"] = (N, 1, L)
    return shapes


def make_state(cfg, dtype=torch.float32):
    return {k: torch.zeros(s, dtype=dtype) for k, s in parameter_shapes(cfg).items()}


CONFIG_SMALL = dict(n_basis=64, kernel_size=16, stride=8, enc_nonlinear=None, sep_hidden_channels=64,
                    sep_bottleneck_channels=32, sep_skip_channels=32, sep_kernel_size=3, sep_num_blocks=2,
                    sep_num_layers=4, mask_nonlinear="sigmoid", causal=False, n_sources=2, eps=EPS)
CONFIG_PAPER = dict(n_basis=512, kernel_size=16, stride=8, enc_nonlinear=None, sep_hidden_channels=512,
                    sep_bottleneck_channels=128, sep_skip_channels=128, sep_kernel_size=3, sep_num_blocks=3,
                    sep_num_layers=8, mask_nonlinear="sigmoid", causal=False, n_sources=2, eps=EPS)
CONFIG_CAUSAL3 = dict(CONFIG_PAPER, causal=True, n_sources=3)


# --------------------------------------------------------------------------
# generic depthwise separable convolution: src/modules/conv.py:13-29
# --------------------------------------------------------------------------
def depthwise_separable_conv1d(x, sd, kernel_size, stride=None, padding=0, dilation=1, prefix=""):
    """DepthwiseSeparableConv1d.forward: src/modules/conv.py:25-29 -- depthwise Conv1d (groups = channels, stride
    defaulting to the kernel size, :17-18) then a pointwise Conv1d, nothing in between."""
    if stride is None:
        stride = kernel_size
    w = sd[prefix + "depthwise_conv1d.weight"]
    y = F.conv1d(x, w, sd.get(prefix + "depthwise_conv1d.bias"), stride=stride, padding=padding, dilation=dilation,
                 groups=w.shape[0])
    return F.conv1d(y, sd[prefix + "pointwise_conv1d.weight"], sd.get(prefix + "pointwise_conv1d.bias"))
