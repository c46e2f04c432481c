"""ORACLE (test infrastructure, never shipped, never measured as the product): CPU fp32 restatement of the
HifiGAN generator forward and of ``HifiGAN.spec2wav``.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Follows (reference file:line):
  * modules/hifigan/hifigan.py:126-142  HifiGanGenerator.forward
  * modules/hifigan/hifigan.py:51-58    ResBlock1.forward
  * modules/hifigan/hifigan.py:108-122  layer construction (channel halving, paddings)
  * modules/hifigan/hifigan.py:144-151  remove_weight_norm  (-> fold_weight_norm below)
  * vocoders/hifigan.py:54-62           spec2wav: [T,80] -> [1,80,T] -> generator -> view(-1)
Pinned against the reference implementation itself by tests/golden/g6_hifigan.npz (oracle/make_golden.py).
"""
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # modules/hifigan/hifigan.py:8


def fold_weight_norm(sd):
    """weight = g * v / ||v||, norm over all dims but 0 (torch.nn.utils.weight_norm, dim=0); applies to
    Conv1d ([out,in,k]) and ConvTranspose1d ([in,out,k]) alike.  Keys without weight_g/_v pass through."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            base = k[: -len(".weight_g")]
            vv = sd[base + ".weight_v"]
            nrm = vv.reshape(vv.shape[0], -1).norm(dim=1).reshape(v.shape)
            out[base + ".weight"] = vv * (v / nrm)
        elif k.endswith(".weight_v"):
            continue
        else:
            out[k] = v
    return out


def _pad(k, d=1):
    return (k * d - d) // 2  # get_padding, hifigan.py:23-24


def resblock1(sd, p, x, k, dilations=(1, 3, 5)):
    for m, d in enumerate(dilations):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, sd[f"{p}.convs1.{m}.weight"], sd[f"{p}.convs1.{m}.bias"], padding=_pad(k, d), dilation=d)
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = F.conv1d(xt, sd[f"{p}.convs2.{m}.weight"], sd[f"{p}.convs2.{m}.bias"], padding=_pad(k, 1))
        x = xt + x
    return x


def generator_forward(sd, cfg, mel, return_stages=False):
    """sd: folded state dict (torch tensors); mel [B,80,T] -> wav [B,1,T*prod(upsample_rates)]"""
    stages = {}
    x = F.conv1d(mel, sd["conv_pre.weight"], sd["conv_pre.bias"], padding=3)
    stages["conv_pre"] = x
    nk = len(cfg["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, sd[f"ups.{i}.weight"], sd[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        stages[f"ups.{i}"] = x
        xs = None
        for j, (rk, rd) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
            r = resblock1(sd, f"resblocks.{i * nk + j}", x, rk, rd)
            xs = r if xs is None else xs + r
        x = xs / nk
        stages[f"stage.{i}"] = x
    x = F.leaky_relu(x)  # default slope 0.01 (hifigan.py:138)
    x = F.conv1d(x, sd["conv_post.weight"], sd["conv_post.bias"], padding=3)
    x = torch.tanh(x)
    return (x, stages) if return_stages else x


def spec2wav(sd, cfg, mel_T80):
    """vocoders/hifigan.py:54-62 for one utterance; mel [T,80] (numpy or tensor) -> 1-D float32 tensor"""
    with torch.no_grad():
        c = torch.as_tensor(mel_T80, dtype=torch.float32).unsqueeze(0).transpose(2, 1)
        return generator_forward(sd, cfg, c).view(-1)
