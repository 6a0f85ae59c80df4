"""CPU tests of the umT5 text-encoder restatement (oracle/t5_oracle.py) against outputs recorded from transformers'
UMT5EncoderModel (tests/golden/t5_tiny.safetensors, made by tests/golden/make_golden_t5.py), and of the host logic of
the drop-in class (key layout, HF key conversion, loud failure without a GPU)."""
import os

import pytest
import torch
from safetensors.torch import load_file

from oracle import t5_oracle as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "t5_tiny.safetensors")
CFG = dict(vocab=300, dim=128, dim_attn=128, dim_ffn=256, num_heads=2, num_layers=2, num_buckets=32, max_distance=128)


@pytest.fixture(scope="module")
def gold():
    d = load_file(GOLD)
    W = {k[2:]: v.float() for k, v in d.items() if k.startswith("w.")}
    return W, d["ids"].long(), d["mask"].long(), d["out"], d["out_nomask"]


def test_oracle_matches_transformers_umt5(gold):
    W, ids, mask, out, out_nomask = gold
    got = T.encode(W, ids, mask, CFG["num_heads"], CFG["num_buckets"], CFG["max_distance"])
    valid = mask.bool()
    assert (got[valid] - out[valid]).abs().max() < 2e-4 * out[valid].abs().max()
    got2 = T.encode(W, ids, None, CFG["num_heads"], CFG["num_buckets"], CFG["max_distance"])
    assert (got2 - out_nomask).abs().max() < 2e-4 * out_nomask.abs().max()
    assert (out[valid] - out_nomask[valid]).abs().max() > 1e-3        # the mask matters on these inputs


def test_relative_position_bucket_matches_transformers():
    d = load_file(GOLD)
    rel, want = d["bucket_rel"].long(), d["bucket_val"].long()
    assert rel.min() <= -512 and rel.max() >= 512
    assert torch.equal(T.relative_position_bucket(rel, CFG["num_buckets"], CFG["max_distance"]), want)


def test_library_bucket_table_matches_transformers():
    """Host-only entry of libvcengine (no GPU work): the table vc_t5_encode uploads is built from this function."""
    from versecrafter_amd import _lib
    lib = _lib.load()
    d = load_file(GOLD)
    for rel, want in zip(d["bucket_rel"].tolist(), d["bucket_val"].tolist()):
        assert lib.vc_t5_relative_bucket(rel, CFG["num_buckets"], CFG["max_distance"]) == want, rel
    assert lib.vc_t5_relative_bucket(0, 3, 128) == -1


def test_state_dict_layout_and_hf_conversion():
    from versecrafter_amd.models import WanT5EncoderModel, convert_hf_umt5_state_dict
    m = WanT5EncoderModel(param_device="meta", **CFG)
    keys = set(m.state_dict().keys())
    d = load_file(GOLD)
    assert keys == {k[2:] for k in d if k.startswith("w.")}
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(d["w." + k].shape), k
    hf = {"shared.weight": torch.zeros(3, 2), "encoder.embed_tokens.weight": torch.zeros(3, 2),
          "encoder.block.1.layer.1.DenseReluDense.wi_0.weight": torch.zeros(1), "encoder.final_layer_norm.weight": torch.zeros(2)}
    assert set(convert_hf_umt5_state_dict(hf)) == {"token_embedding.weight", "blocks.1.ffn.gate.0.weight", "norm.weight"}
    # full-size key count of umT5-XXL (wan_civitai.yaml:14-26): 24 layers x 10 + 2
    big = WanT5EncoderModel(param_device="meta")
    assert len(big.state_dict()) == 242


def test_text_encoder_has_no_cpu_path(gold):
    from versecrafter_amd.models import WanT5EncoderModel
    W, ids, mask, *_ = gold
    m = WanT5EncoderModel(**CFG)
    m.load_state_dict({k: v.bfloat16() for k, v in W.items()})
    with pytest.raises(RuntimeError):
        m(ids, attention_mask=mask)
    with pytest.raises(NotImplementedError):
        WanT5EncoderModel(dim=128, dim_attn=96, num_heads=2, param_device="meta")


@pytest.mark.parametrize("fmt", ["pth", "safetensors", "hf-safetensors"])
def test_text_encoder_from_pretrained_formats(tmp_path, gold, fmt):
    """CLI.py:243-247: a checkpoint FILE in the upstream key layout (.pth loaded with weights_only=True, or .safetensors),
    or in the transformers umT5 layout; yaml-style kwargs incl. the path keys the CLI passes along."""
    from safetensors.torch import save_file
    from versecrafter_amd.models import WanT5EncoderModel
    W = {k: v.bfloat16().contiguous() for k, v in gold[0].items()}
    if fmt == "pth":
        path = str(tmp_path / "t5.pth")
        torch.save(W, path)
    elif fmt == "safetensors":
        path = str(tmp_path / "t5.safetensors")
        save_file(W, path)
    else:
        inv = {"norm1.weight": "layer.0.layer_norm.weight", "attn.q.weight": "layer.0.SelfAttention.q.weight",
               "attn.k.weight": "layer.0.SelfAttention.k.weight", "attn.v.weight": "layer.0.SelfAttention.v.weight",
               "attn.o.weight": "layer.0.SelfAttention.o.weight",
               "pos_embedding.embedding.weight": "layer.0.SelfAttention.relative_attention_bias.weight",
               "norm2.weight": "layer.1.layer_norm.weight", "ffn.gate.0.weight": "layer.1.DenseReluDense.wi_0.weight",
               "ffn.fc1.weight": "layer.1.DenseReluDense.wi_1.weight", "ffn.fc2.weight": "layer.1.DenseReluDense.wo.weight"}
        hf = {"shared.weight": W["token_embedding.weight"], "encoder.final_layer_norm.weight": W["norm.weight"]}
        for k, v in W.items():
            if k.startswith("blocks."):
                _, n, rest = k.split(".", 2)
                hf[f"encoder.block.{n}.{inv[rest]}"] = v
        path = str(tmp_path / "hf.safetensors")
        save_file(hf, path)
    kw = dict(CFG, text_encoder_subpath="x", tokenizer_subpath="y", text_length=512, shared_pos=False, dropout=0.0)
    m = WanT5EncoderModel.from_pretrained(path, additional_kwargs=kw, torch_dtype=torch.bfloat16)
    assert m.dtype == torch.bfloat16
    for k, v in m.state_dict().items():
        assert torch.equal(v, W[k]), k
    with pytest.raises(RuntimeError):
        WanT5EncoderModel.from_pretrained(str(tmp_path / "missing.pth"), additional_kwargs=kw)
