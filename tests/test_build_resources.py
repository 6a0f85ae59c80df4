"""Compile-time resource check (no GPU): the kernels a cross-attention call can reach -- the T5 path of wan_transformer3d.py:425-430
for any prompt of <= 512 tokens -- must not spill (round-2 verdict: the <MERGE, 4-wave> pipelined instantiation spilled 893 VGPRs
and was what a 300-token prompt ran; it is gone, attention_stream.hip took its place)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "versecrafter_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def _resources(src):
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c", os.path.join(CSRC, src), "-o",
                        "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out, name = {}, None
    for ln in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            name = m.group(1)
            out[name] = {}
        m = re.search(r"remark:\s+(VGPRs Spill|ScratchSize \[bytes/lane\]|VGPRs): (\d+)", ln)
        if m and name:
            out[name][m.group(1)] = int(m.group(2))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_cross_attention_kernels_do_not_spill():
    res = _resources("attention_stream.hip")
    assert len(res) == 2 and all("attn_stream_kernel" in k for k in res), res
    for k, v in res.items():
        assert v["VGPRs Spill"] == 0 and v["ScratchSize [bytes/lane]"] == 0, (k, v)
    res = _resources("attention.hip")
    short = {k: v for k, v in res.items() if "attn_short_kernel" in k}
    assert len(short) == 2
    for k, v in short.items():
        assert v["VGPRs Spill"] == 0 and v["ScratchSize [bytes/lane]"] == 0, (k, v)
    # the software-pipelined kernel is instantiated for self-attention only: {plain, segmented} x {4, 8 waves}, never with the
    # padded-key folding (template arguments <SEG, NW, MERGE>: the MERGE = true instantiations are gone)
    pipe = [k for k in res if "attn_fwd_pipe_kernel" in k]
    assert len(pipe) == 4 and all(k.endswith("ELb0EEEv12VcAttnParamsii") for k in pipe), pipe


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_fp8_attention_kernels_have_no_scratch():
    """attn_fp8_kernel issues its MFMAs by inline asm (the builtin leaves the accumulators untied and spills them).  hipcc then knows
    nothing of their latency: a spill store of an accumulator placed behind such a statement reads the registers before the matrix pipe
    has written them.  Round 4 found exactly that in specialised tail instances (wrong results for short key sequences); the kernel now
    has one body and must stay free of scratch: this is a CORRECTNESS condition, not a performance one."""
    res = _resources("attention_fp8.hip")
    att = {k: v for k, v in res.items() if "attn_fp8_kernel" in k}
    assert len(att) == 2, res.keys()
    for k, v in att.items():
        assert v["VGPRs Spill"] == 0 and v["ScratchSize [bytes/lane]"] == 0 and v["VGPRs"] <= 256, (k, v)
