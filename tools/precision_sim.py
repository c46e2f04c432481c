"""CPU emulation of the vocoder's matrix-core arithmetic options (test tooling; imports the oracle, never the product).

Every convolution of the HifiGAN generator is evaluated in fp32 on operands rounded the way a candidate MFMA scheme
would round them (fp32 accumulation is common to all schemes), and the waveform is compared with the exact fp32 oracle:
RMS(w - ref), |RMS(w) - RMS(ref)|.  `--by-stage` gives the per-stage attribution VERDICT r01 asks for (one stage
rounded, everything else exact).

schemes: name = <activation term list>/<weight term list>, e.g.  b/b  (bf16 single),  h/h  (fp16 single),
         hh/h (activation hi+lo fp16, weight fp16: 2 MFMAs), hh/hh3 (3-term), bb/bb3 (bf16x3) ...
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dict_tts_amd import synth  # noqa: E402
from oracle import hifigan_ref as href  # noqa: E402


def rnd(x, kind):
    if kind == "b":
        return x.to(torch.bfloat16).to(torch.float32)
    if kind == "h":
        return x.to(torch.float16).to(torch.float32)
    raise ValueError(kind)


def split(x, kinds):
    """list of terms whose sum approximates x: first term = round(x), second = round(x - first) ..."""
    out, r = [], x
    for k in kinds:
        t = rnd(r, k)
        out.append(t)
        r = r - t
    return out


class Scheme:
    def __init__(self, spec):
        # spec "XX/WW[n]"; n = number of product terms kept (default all pairs with i+j < max(len))
        a, w = spec.split("/")
        self.nterms = None
        if w[-1].isdigit():
            self.nterms = int(w[-1])
            w = w[:-1]
        self.a, self.w = a, w
        self.exact = spec == "f/f"

    def pairs(self):
        p = [(i, j) for i in range(len(self.a)) for j in range(len(self.w))]
        p.sort(key=lambda ij: (ij[0] + ij[1], ij[0]))
        if self.nterms is not None:
            p = p[: self.nterms]
        else:
            p = [ij for ij in p if ij[0] + ij[1] < max(len(self.a), len(self.w))]
        return p

    def conv(self, fn, x, w, b, **kw):
        if self.exact:
            return fn(x, w, b, **kw)
        xs, ws = split(x, self.a), split(w, self.w)
        y = None
        for i, j in self.pairs():
            t = fn(xs[i], ws[j], None, **kw)
            y = t if y is None else y + t
        return y + b.view(1, -1, 1)


def generator(sd, cfg, mel, scheme_of, stream=None):
    """scheme_of(layer_name) -> Scheme.  stream: None, or (set of (stage, kernel_size), set of iteration indices, kind): the residual
    stream r LEAVING those iterations of those ResBlocks is stored in a 16-bit type (vpair's inter-iteration stream, round 6)."""
    L = href.LRELU_SLOPE
    x = scheme_of("conv_pre").conv(F.conv1d, mel, sd["conv_pre.weight"], sd["conv_pre.bias"], padding=3)
    nk = len(cfg["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, L)
        x = scheme_of(f"ups.{i}").conv(F.conv_transpose1d, x, sd[f"ups.{i}.weight"], sd[f"ups.{i}.bias"], stride=u,
                                       padding=(k - u) // 2)
        xs = None
        for j, (rk, rd) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
            p = f"resblocks.{i * nk + j}"
            sc = scheme_of(f"stage.{i}")
            r = x
            for m, d in enumerate(rd):
                xt = F.leaky_relu(r, L)
                xt = sc.conv(F.conv1d, xt, sd[f"{p}.convs1.{m}.weight"], sd[f"{p}.convs1.{m}.bias"],
                             padding=href._pad(rk, d), dilation=d)
                xt = F.leaky_relu(xt, L)
                xt = sc.conv(F.conv1d, xt, sd[f"{p}.convs2.{m}.weight"], sd[f"{p}.convs2.{m}.bias"], padding=href._pad(rk, 1))
                r = xt + r
                if stream is not None and (i, rk) in stream[0] and m in stream[1]:
                    r = rnd(r, stream[2])
            xs = r if xs is None else xs + r
        x = xs / nk
    x = F.leaky_relu(x)
    x = scheme_of("conv_post").conv(F.conv1d, x, sd["conv_post.weight"], sd["conv_post.bias"], padding=3)
    return torch.tanh(x)


def rms(a):
    return float(np.sqrt(np.mean(np.square(np.asarray(a, np.float64)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--schemes", default="b/b,h/h,hh/h,h/hh,hh/hh,bb/bb,bb/b,b/bb")
    ap.add_argument("--by-stage", default="")
    ap.add_argument("--mix", default="", help="layer=scheme,... with 'default=scheme'")
    ap.add_argument("--seeds", type=int, default=2)
    ap.add_argument("--stream", default="", help="stage:kernel,... whose inter-iteration residual stream is stored 16-bit, e.g. 0:7,0:11,1:7,1:11 "
                    "('all' = every ResBlock); evaluated on top of --stream-scheme")
    ap.add_argument("--stream-iters", default="0,1", help="iterations whose OUTPUT is stored 16-bit (2 = also the ResBlock's result)")
    ap.add_argument("--stream-kind", default="h", help="h = fp16, b = bf16")
    ap.add_argument("--stream-scheme", default="h/h", help="arithmetic of the ResBlock stages under --stream (serial convolutions exact, as bf16x3 is)")
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    cfg = synth.hifigan_config()
    sd = href.fold_weight_norm({k: torch.from_numpy(v) for k, v in synth.hifigan_state_dict(1234).items()})
    exact = Scheme("f/f")
    layers = ["conv_pre"] + [f"ups.{i}" for i in range(4)] + [f"stage.{i}" for i in range(4)] + ["conv_post"]
    with torch.no_grad():
        for s in range(a.seeds):
            mel = torch.from_numpy(synth.random_mel(1234 + s, a.frames, f"psim{s}")).T.unsqueeze(0).contiguous()
            ref = generator(sd, cfg, mel, lambda n: exact).numpy().ravel()
            print(f"seed {s}: frames {a.frames} rms(ref) {rms(ref):.4f}")
            for spec in a.schemes.split(","):
                if not spec:
                    continue
                sc = Scheme(spec)
                w = generator(sd, cfg, mel, lambda n: sc).numpy().ravel()
                print(f"  {spec:8s} terms {len(sc.pairs())}  rms(d) {rms(w - ref):.3e}  |drms| {abs(rms(w) - rms(ref)):.2e}  "
                      f"max {np.abs(w - ref).max():.2e}")
            if a.by_stage:
                sc = Scheme(a.by_stage)
                for ln in layers:
                    w = generator(sd, cfg, mel, lambda n: sc if n == ln else exact).numpy().ravel()
                    print(f"  only {ln:10s} in {a.by_stage}: rms(d) {rms(w - ref):.3e}")
            if a.stream:
                nk = cfg["resblock_kernel_sizes"]
                pairs = {(i, k) for i in range(4) for k in nk} if a.stream == "all" else {tuple(int(v) for v in t.split(":")) for t in a.stream.split(",")}
                iters = {int(v) for v in a.stream_iters.split(",")}
                sc = Scheme(a.stream_scheme)
                pick = lambda n: sc if n.startswith("stage.") else exact
                w0 = generator(sd, cfg, mel, pick).numpy().ravel()
                w1 = generator(sd, cfg, mel, pick, stream=(pairs, iters, a.stream_kind)).numpy().ravel()
                print(f"  stages {a.stream_scheme}, fp32 stream: rms(d) {rms(w0 - ref):.3e}   |  {a.stream_kind}16 stream after iterations {sorted(iters)} of {sorted(pairs)}: "
                      f"rms(d) {rms(w1 - ref):.3e} |drms| {abs(rms(w1) - rms(ref)):.2e}")
            if a.mix:
                m = dict(kv.split("=") for kv in a.mix.split(","))
                scs = {k: Scheme(v) for k, v in m.items()}
                w = generator(sd, cfg, mel, lambda n: scs.get(n, scs["default"])).numpy().ravel()
                print(f"  mix {a.mix}: rms(d) {rms(w - ref):.3e} |drms| {abs(rms(w) - rms(ref)):.2e}")


if __name__ == "__main__":
    main()
