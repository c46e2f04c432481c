"""ORACLE (test infrastructure, never shipped, never measured as the product): CPU fp32 restatement of
``PortaSpeech_dict.forward(infer=True)`` — rows A1..A10 of SURVEY.md §8a.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Every function cites the reference file:line it restates.  State-dict key names are the reference's
(``state_dict['model']``); weight-norm pairs must be folded first (oracle/hifigan_ref.py:fold_weight_norm,
which restates tasks/tts/ps_flow.py:262-268).  Pinned against the reference implementation itself (imported
in the build container) by tests/golden/g*.npz — see oracle/make_golden.py.
"""
import math

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------------
# A2: transformer ("FFT") encoder blocks — modules/commons/rel_transformer_encoder.py
# ---------------------------------------------------------------------------------------------------------
def layer_norm_c(x, gamma, beta, eps=1e-4):
    """LayerNorm over the channel dim of [B,C,T] (rel_transformer_encoder.py:261-279)"""
    mean = torch.mean(x, 1, keepdim=True)
    var = torch.mean((x - mean) ** 2, 1, keepdim=True)
    x = (x - mean) * torch.rsqrt(var + eps)
    return x * gamma.view(1, -1, 1) + beta.view(1, -1, 1)


def mha(sd, p, x, attn_mask, n_heads):
    """MultiHeadAttention.forward / .attention with window_size=None (rel_transformer_encoder.py:117-158)"""
    q = F.conv1d(x, sd[p + ".conv_q.weight"], sd[p + ".conv_q.bias"])
    k = F.conv1d(x, sd[p + ".conv_k.weight"], sd[p + ".conv_k.bias"])
    v = F.conv1d(x, sd[p + ".conv_v.weight"], sd[p + ".conv_v.bias"])
    b, d, t = q.shape
    kc = d // n_heads
    q = q.view(b, n_heads, kc, t).transpose(2, 3)
    k = k.view(b, n_heads, kc, t).transpose(2, 3)
    v = v.view(b, n_heads, kc, t).transpose(2, 3)
    scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(kc)
    scores = scores.masked_fill(attn_mask == 0, -1e4)
    p_attn = F.softmax(scores, dim=-1)
    out = torch.matmul(p_attn, v)
    out = out.transpose(2, 3).contiguous().view(b, d, t)
    return F.conv1d(out, sd[p + ".conv_o.weight"], sd[p + ".conv_o.bias"])


def ffn(sd, p, x, x_mask, k):
    """FFN.forward, activation=None -> ReLU (rel_transformer_encoder.py:250-258)"""
    x = F.conv1d(x * x_mask, sd[p + ".conv_1.weight"], sd[p + ".conv_1.bias"], padding=k // 2)
    x = torch.relu(x)
    x = F.conv1d(x * x_mask, sd[p + ".conv_2.weight"], sd[p + ".conv_2.bias"])
    return x * x_mask


def rel_encoder(sd, p, x, x_mask, n_layers=4, n_heads=2, k=5):
    """Encoder.forward with pre_ln=True (rel_transformer_encoder.py:55-79)"""
    attn_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
    for i in range(n_layers):
        x = x * x_mask
        x_ = x
        x = layer_norm_c(x, sd[f"{p}.norm_layers_1.{i}.gamma"], sd[f"{p}.norm_layers_1.{i}.beta"])
        x = x_ + mha(sd, f"{p}.attn_layers.{i}", x, attn_mask, n_heads)
        x_ = x
        x = layer_norm_c(x, sd[f"{p}.norm_layers_2.{i}.gamma"], sd[f"{p}.norm_layers_2.{i}.beta"])
        x = x_ + ffn(sd, f"{p}.ffn_layers.{i}", x, x_mask, k)
    x = layer_norm_c(x, sd[f"{p}.last_ln.gamma"], sd[f"{p}.last_ln.beta"])
    return x * x_mask


# ---------------------------------------------------------------------------------------------------------
# A3: S2PA dictionary attention — modules/dict_tts/layers/dict_encoder.py:32-66, layers/utils.py
# ---------------------------------------------------------------------------------------------------------
def s2pa_attention(sd, p, x, keys, values, key_map, pinyin, pinyin_map, pron_modified, language="zh"):
    """x [B,192,T_w]; keys/values [B,T_w,L_k,768]; key_map [B,T_w,L_k] f32; pinyin/pinyin_map [B,T_w,P] i64;
    pron_modified [B,T_w] i64 -> context [B,192,T_w], align [B,1,L_k,T_w], pron [B,192,T_w],
    pron_weights [B,T_w,P]."""
    key_size = keys.shape[-1]
    q = F.linear(x.transpose(1, 2), sd[p + ".q_transform.weight"])          # [B,T,192]
    k = F.linear(keys, sd[p + ".k_transform.weight"])                        # [B,T,L,192]
    v = F.linear(values, sd[p + ".v_transform.weight"])
    q = q * key_size ** -0.5                                                 # dict_encoder.py:45-46 (num_heads=1)
    logits = torch.matmul(k, q.unsqueeze(-1)).squeeze(-1)                    # [B,T,L]
    logits = logits.masked_fill(key_map == 0, -1e9)                          # utils.py:40-47 mask_logits
    weights = F.softmax(logits, dim=-1)
    align = weights.unsqueeze(1).permute(0, 1, 3, 2)                         # [B,1,L,T]
    context = torch.matmul(weights.unsqueeze(-2), v).squeeze(-2)             # [B,T,192]
    context = F.linear(context, sd[p + ".output_transform.weight"]).transpose(1, 2)
    # pronunciation branch: utils.py:49-58 mask_weights_attn
    pin = F.embedding(pinyin, sd[p + ".pinyin_embedding.weight"])            # [B,T,P,192]
    res = torch.zeros(weights.size(0), weights.size(1), pin.size(2))
    for i in range(1, int(key_map.max()) + 1):
        merge = (weights * key_map.eq(i).float()).sum(dim=-1, keepdim=True)
        res = res + merge * pinyin_map.eq(i).float()
    if language == "zh":                                                     # utils.py:109-115 add_pron_rule
        forced = res.clone()
        for i in range(1, int(pinyin_map.max()) + 1):
            sel = pron_modified == i
            forced[sel] = (pinyin_map[sel] == i).float()
        res = forced - res + res  # the straight-through form of utils.py:114, kept for its fp32 rounding
    pron = torch.matmul(res.unsqueeze(-2), pin).squeeze(-2).transpose(1, 2)
    return context, align, pron, res


def dict_encoder(sd, word_tokens, dict_msg, pron_modified, hidden=192, n_heads=2, ffn_k=5):
    """S2PATextEncoder.forward + DictEncoder.forward (dict_encoder.py:130-144,165-171) -> word_encoder_out
    [B,T_w,192], dict_attn, pron_attn [B,T_w,P], context [B,T_w,192]"""
    p = "dict_encoder.S2PA_module"
    keys, values, key_map, pinyin, pinyin_map = dict_msg
    x_lengths = (word_tokens > 0).long().sum(-1)
    x = F.embedding(word_tokens, sd[p + ".word_emb.weight"]) * math.sqrt(hidden)
    x = x.transpose(1, -1)
    T = x.size(2)
    x_mask = (torch.arange(T).unsqueeze(0) < x_lengths.unsqueeze(1)).unsqueeze(1).to(x.dtype)  # sequence_mask
    x = rel_encoder(sd, p + ".semantic_encoder", x, x_mask, 4, n_heads, ffn_k)
    context, dict_attn, pron, pron_align = s2pa_attention(sd, p + ".s2pa_attention", x, keys, values, key_map,
                                                          pinyin, pinyin_map, pron_modified)
    context = context * x_mask
    x = context + pron
    x = rel_encoder(sd, p + ".linguistic_encoder", x, x_mask, 4, n_heads, ffn_k)
    x = x.transpose(1, 2) * (word_tokens > 0).float()[:, :, None]
    return x, dict_attn, pron_align, context.transpose(1, 2)


# ---------------------------------------------------------------------------------------------------------
# A5-A7: duration predictor, length regulator, expansion
# ---------------------------------------------------------------------------------------------------------
def duration_predictor(sd, xs, x_masks, n_layers=3, k=5):
    """DurationPredictor.forward, padding='SAME' (modules/portaspeech/model.py:58-66) with the torch
    LayerNorm(eps=1e-5) over channels (modules/fastspeech/tts_modules.py:60-79)"""
    xs = xs.transpose(1, -1)
    keep = (1 - x_masks.float())[:, None, :]
    for i in range(n_layers):
        xs = F.pad(xs, ((k - 1) // 2, (k - 1) // 2))
        xs = F.conv1d(xs, sd[f"dur_predictor.conv.{i}.1.weight"], sd[f"dur_predictor.conv.{i}.1.bias"])
        xs = torch.relu(xs)
        xs = F.layer_norm(xs.transpose(1, -1), (xs.shape[1],), sd[f"dur_predictor.conv.{i}.3.weight"],
                          sd[f"dur_predictor.conv.{i}.3.bias"], 1e-5).transpose(1, -1)
        xs = xs * keep
    xs = F.linear(xs.transpose(1, -1), sd["dur_predictor.linear.0.weight"], sd["dur_predictor.linear.0.bias"])
    xs = F.softplus(xs)[:, :, 0]
    return xs * (1 - x_masks.float())


def length_regulator(dur, ilens):
    """LengthRegulator.forward(alpha=1) + pad_list (modules/fastspeech/tts_modules.py:171-251): integer
    durations [B,T_w], valid lengths [B] -> mel2word [B,T_mel'] (1-based word index, 0 = padding)"""
    rows = []
    for d, n in zip(dur.tolist(), ilens.tolist()):
        d = d[:n]
        if sum(d) == 0:          # "all of the predicted durations are 0. fill 0 with 1." (:248-250)
            d = [1] * len(d)
        r = []
        for idx, dd in enumerate(d):
            r += [idx + 1] * int(dd)
        rows.append(r)
    T = max(len(r) for r in rows)
    out = torch.zeros(len(rows), T, dtype=torch.long)
    for b, r in enumerate(rows):
        out[b, :len(r)] = torch.tensor(r, dtype=torch.long)
    return out


def add_dur(sd, dur_input, mel2word):
    """PortaSpeech_dict.add_dur (modules/dict_tts/model.py:64-82), dur_scale='log'"""
    src_padding = dur_input.abs().sum(-1) == 0
    dur = duration_predictor(sd, dur_input, src_padding)
    if mel2word is None:
        d = torch.clamp(torch.round(dur.exp() - 1), min=0).long()
        mel2word = length_regulator(d, (1 - src_padding.long()).sum(-1))
    return dur, mel2word


def expand(word_encoder_out, mel2word, frames_multiple=4):
    """run_text_encoder tail (modules/dict_tts/model.py:98-107)"""
    if mel2word.shape[1] % frames_multiple > 0:
        pad_len = frames_multiple - mel2word.shape[1] % frames_multiple
        mel2word = torch.cat([mel2word] + [mel2word[:, -1:]] * pad_len, -1)
    tgt_nonpadding = (mel2word > 0).float()[:, :, None]
    x = F.pad(word_encoder_out, [0, 0, 1, 0])
    x = torch.gather(x, 1, mel2word[..., None].repeat([1, 1, x.shape[-1]]))
    return x, tgt_nonpadding, mel2word


# ---------------------------------------------------------------------------------------------------------
# A8-A10: FVAE prior flow + decoder
# ---------------------------------------------------------------------------------------------------------
def wn(sd, p, x, g, hidden, k, n_layers):
    """WN.forward with x_mask = 1, dilation_rate 1 (modules/commons/wavenet.py:54-78, :5-11)"""
    output = torch.zeros_like(x)
    g = F.conv1d(g, sd[p + ".cond_layer.weight"], sd[p + ".cond_layer.bias"])
    for i in range(n_layers):
        x_in = F.conv1d(x, sd[f"{p}.in_layers.{i}.weight"], sd[f"{p}.in_layers.{i}.bias"], padding=(k - 1) // 2)
        in_act = x_in + g[:, i * 2 * hidden:(i + 1) * 2 * hidden, :]
        acts = torch.tanh(in_act[:, :hidden, :]) * torch.sigmoid(in_act[:, hidden:, :])
        rs = F.conv1d(acts, sd[f"{p}.res_skip_layers.{i}.weight"], sd[f"{p}.res_skip_layers.{i}.bias"])
        if i < n_layers - 1:
            x = x + rs[:, :hidden, :]
            output = output + rs[:, hidden:, :]
        else:
            output = output + rs
    return output


def prior_flow_reverse(sd, z, g_sqz, n_flows=4, hidden=64, k=3, n_layers=4):
    """ResidualCouplingBlock.forward(reverse=True) (modules/portaspeech/glow_modules.py:157-163): for flow
    in reversed([rcl0, flip, rcl1, flip, ...]); ResidualCouplingLayer.forward mean_only (:108-128); Flip (:9-13)"""
    half = z.shape[1] // 2
    for f in reversed(range(n_flows)):
        z = torch.flip(z, [1])
        p = f"fvae.prior_flow.flows.{2 * f}"
        x0, x1 = z[:, :half], z[:, half:]
        h = F.conv1d(x0, sd[p + ".pre.weight"], sd[p + ".pre.bias"])
        h = wn(sd, p + ".enc", h, g_sqz, hidden, k, n_layers)
        m = F.conv1d(h, sd[p + ".post.weight"], sd[p + ".post.bias"])
        x1 = (x1 - m) * torch.exp(-torch.zeros_like(m))
        z = torch.cat([x0, x1], 1)
    return z


def fvae_infer(sd, g, z_p, hidden=192, k=5, n_layers=4):
    """FVAE_semantics.forward(infer=True) with semantics = 0 and an explicit prior sample
    (modules/dict_tts/fvae_semantics.py:84-115, :53-58); g [B,192,T] -> mel [B,80,T]"""
    g_sqz = F.conv1d(g, sd["fvae.g_pre_net.0.weight"], sd["fvae.g_pre_net.0.bias"], stride=4, padding=2)
    z = prior_flow_reverse(sd, z_p, g_sqz)
    x = F.conv_transpose1d(z, sd["fvae.decoder.pre_net.0.weight"], sd["fvae.decoder.pre_net.0.bias"], stride=4)
    x = wn(sd, "fvae.decoder.wn", x, g, hidden, k, n_layers)
    return F.conv1d(x, sd["fvae.decoder.out_proj.weight"], sd["fvae.decoder.out_proj.bias"]), z


# ---------------------------------------------------------------------------------------------------------
# whole model
# ---------------------------------------------------------------------------------------------------------
def forward_infer(sd, word_tokens, dict_msg, pron_modified, mel2word=None, z_p=None):
    """PortaSpeech_dict.forward(infer=True), no speaker embedding, no post-glow
    (modules/dict_tts/model.py:36-62,84-121).  z_p: explicit [B,16,T_mel/4] prior sample or a callable
    (B, T4) -> tensor (the reference draws it from the CPU global RNG, fvae_semantics.py:110-111)."""
    with torch.no_grad():
        ret = {}
        padding_mask = word_tokens.eq(0)
        nonpadding = (1 - padding_mask.float())[:, :, None]
        weo, dict_attn, pron_attn, context = dict_encoder(sd, word_tokens, dict_msg, pron_modified)
        ret.update(dict_attn=dict_attn, pron_attn=pron_attn, word_encoder_out=weo, context=context)
        dur, mel2word = add_dur(sd, weo * nonpadding, mel2word)
        ret["dur"] = dur
        x, tgt_nonpadding, mel2word = expand(weo, mel2word)
        ret["mel2word"] = mel2word
        x = x * tgt_nonpadding
        ret["x_mask"] = tgt_nonpadding
        ret["decoder_inp"] = x
        g = x.transpose(1, 2)
        if callable(z_p):
            z_p = z_p(g.shape[0], g.shape[2] // 4)
        mel, z = fvae_infer(sd, g, z_p)
        ret["z_p"] = z
        ret["mel_out"] = ret["mel_out_fvae"] = mel.transpose(1, 2)
        return ret


def decode_pinyin(pron_attn, pinyin):
    """after_infer's pinyin decode (tasks/tts/dict_tts.py:294-304) for ONE utterance: pron_attn [T_w,P],
    pinyin [T_w,P] -> list of pinyin token ids (two per inner word)"""
    max_idx = pron_attn.max(dim=-1)[1]
    out = []
    for i in range(1, pinyin.shape[0] - 1):
        out += pinyin[i][max_idx[i]:max_idx[i] + 2].tolist()
    return out
