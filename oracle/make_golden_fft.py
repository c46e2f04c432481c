#!/usr/bin/env python3
"""Generate tests/golden/g8_fft_blocks.npz by running the REFERENCE's FFTBlocks (imported from /root/reference,
modules/fastspeech/tts_modules.py:458-523) on the inputs of tests/golden_cases.py:g8_inputs with the weights of
dict_tts_amd/synth.py:fft_blocks_state_dict.  TEST INFRASTRUCTURE; runs only in the build container.  Same import
recipe as oracle/make_golden.py (stub finder, cwd = reference root, Biaobei Dict-TTS hparams: hidden 192, 2 heads,
ffn_act gelu, ffn_padding SAME).  Nothing of the reference is copied; only its outputs are stored."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (the stub finder and the paths)


def main():
    sys.dont_write_bytecode = True
    sys.meta_path.insert(0, mg._Finder())
    sys.path.insert(0, mg.REPO)
    sys.path.insert(0, os.path.join(mg.REPO, "tests"))
    sys.path.insert(0, mg.REF)
    os.chdir(mg.REF)
    import warnings
    warnings.filterwarnings("ignore")
    import numpy as np
    import torch
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from dict_tts_amd import synth
    import golden_cases as gc
    from utils.hparams import set_hparams
    set_hparams(config="egs/datasets/audio/biaobei/dict_tts.yaml", exp_name="",
                hparams_str="use_word_input=True,word_size=8000,use_dict=True", print_hparams=False)
    from modules.fastspeech.tts_modules import FFTBlocks
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    out = {}
    for name, cfg in gc.G8_CASES.items():
        m = FFTBlocks(192, cfg["layers"], ffn_kernel_size=cfg["kernel_size"], num_heads=2,
                      use_pos_embed=cfg["use_pos_embed"], use_last_norm=cfg["use_last_norm"])
        sd = {k: T(v) for k, v in synth.fft_blocks_state_dict(gc.SEED, 192, **cfg).items()}
        print(name, "load_state_dict(strict=True):", m.load_state_dict(sd, strict=True))
        m.eval()
        x, lens = gc.g8_inputs(name)
        with torch.no_grad():
            y = m(T(x))                                      # padding mask derived from the values, as callers do
            hid = m(T(x), return_hiddens=True)               # [L,B,T,C] per-layer outputs
        out[name + ".out"] = y.numpy()
        out[name + ".hidden0"] = hid[0].numpy()
        print(name, tuple(y.shape), "rms", float(y.pow(2).mean().sqrt()), "absmax", float(y.abs().max()))
    np.savez_compressed(os.path.join(mg.OUT, "g8_fft_blocks.npz"), **out)
    print("g8_fft_blocks.npz", os.path.getsize(os.path.join(mg.OUT, "g8_fft_blocks.npz")))


if __name__ == "__main__":
    main()
