#!/usr/bin/env python3
"""Records outputs of transformers' UMT5EncoderModel (an independent implementation of the umT5 encoder; the reference's
own WanT5EncoderModel is un-vendored) on seeded inputs -> tests/golden/t5_tiny.safetensors.  Run in the build container:
    python tests/golden/make_golden_t5.py
Weights are rounded to bf16 before the run so that a bf16 engine sees exactly the recorded values."""
import os
import sys

import torch
from safetensors.torch import save_file

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from transformers import UMT5Config, UMT5EncoderModel   # noqa: E402

from versecrafter_amd.models.wan_text_encoder import convert_hf_umt5_state_dict   # noqa: E402

CFG = dict(vocab=300, dim=128, dim_attn=128, dim_ffn=256, num_heads=2, num_layers=2, num_buckets=32, max_distance=128)


def main():
    torch.manual_seed(1234)
    cfg = UMT5Config(vocab_size=CFG["vocab"], d_model=CFG["dim"], d_kv=CFG["dim_attn"] // CFG["num_heads"], d_ff=CFG["dim_ffn"],
                     num_layers=CFG["num_layers"], num_heads=CFG["num_heads"],
                     relative_attention_num_buckets=CFG["num_buckets"], relative_attention_max_distance=CFG["max_distance"],
                     dropout_rate=0.0, feed_forward_proj="gated-gelu", layer_norm_epsilon=1e-6)
    m = UMT5EncoderModel(cfg).eval()
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for n, p in m.named_parameters():            # non-trivial norms / position biases, bf16-exact values
            if p.dim() == 1:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif "relative_attention_bias" in n:
                p.copy_(torch.randn(p.shape, generator=g))
            elif "embed_tokens" in n or n == "shared.weight":
                p.copy_(torch.randn(p.shape, generator=g))
            else:
                scale = p.shape[1] ** -0.5 * (0.35 if (".q." in n or ".k." in n) else 1.0)
                p.copy_(torch.randn(p.shape, generator=g) * scale)
            p.copy_(p.bfloat16().float())
    B, L = 2, 128
    ids = torch.randint(0, CFG["vocab"], (B, L), generator=g)
    mask = torch.ones(B, L, dtype=torch.long)
    mask[0, 77:] = 0
    mask[1, 100:] = 0
    with torch.no_grad():
        out = m(ids, attention_mask=mask)[0]
        out_nomask = m(ids)[0]
    sd = {k: v.clone().contiguous() for k, v in convert_hf_umt5_state_dict(m.state_dict()).items()}
    sd = {"w." + k: v.bfloat16() for k, v in sd.items()}          # bf16-exact by construction
    rel = torch.arange(-600, 601)
    att = m.encoder.block[0].layer[0].SelfAttention
    bucket = att._relative_position_bucket(rel)            # transformers' own bucket function (bidirectional encoder)
    sd.update(ids=ids.to(torch.int32), mask=mask.to(torch.int32), out=out.contiguous(), out_nomask=out_nomask.contiguous(),
              bucket_rel=rel.to(torch.int32), bucket_val=bucket.to(torch.int32))
    path = os.path.join(ROOT, "tests", "golden", "t5_tiny.safetensors")
    save_file(sd, path)
    print("wrote", path, {k: tuple(v.shape) for k, v in sd.items() if not k.startswith("w.")})


if __name__ == "__main__":
    main()
