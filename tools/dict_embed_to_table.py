#!/usr/bin/env python3
"""One-time conversion of the reference's dictionary dataset into the resident-table arrays:

    python tools/dict_embed_to_table.py <binary_data_dir>/dict_embed <binary_data_dir>/pinyin_encoder.pkl table.npz

table.npz holds tok_off / keys / values (absent when equal to keys) / key_map / pin_off / pinyin / pinyin_map — exactly the
arguments of dtts_dict_table_upload; ``PortaSpeech_dict.upload_dict_table(dict(np.load('table.npz')))`` makes it resident."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dict_tts_amd.dict_embed import table_from_dict_embed  # noqa: E402


def main():
    if len(sys.argv) != 4:
        print(__doc__)
        return 2
    t = table_from_dict_embed(sys.argv[1], sys.argv[2])
    arrays = {k: v for k, v in t.items() if k not in ("ids", "L", "P") and v is not None}
    np.savez(sys.argv[3], **arrays)
    print(f"{len(t['L'])} entries, {int(t['tok_off'][-1])} gloss tokens ({t['keys'].nbytes / 2**20:.1f} MiB of keys), "
          f"L_max {int(t['L'].max())}, P_max {int(t['P'].max())}, values {'== keys' if t['values'] is None else 'separate'}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
