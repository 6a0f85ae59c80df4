// One denoise-step forward of VerseCrafterWanTransformer3DModel through the C ABI ALONE (include/vcengine.h):
// no Python, no torch -- what a C/C++ host (or any FFI) does to use libvcengine.  This is the binding sequence of
// INTEGRATION.md in compiled form:
//     vc_create -> vc_load_weight (every state-dict key) -> vc_set_rope_table -> vc_prepare_video -> vc_forward
//
//   c_abi_denoise_step <bundle.bin> <out.bin>
// bundle.bin (written by tests/test_gpu_c_abi.py): a flat little-endian container
//     u32 magic 'VCB1', u32 n_records, then per record: u32 name_len, name, u32 kind (0 bf16, 1 f32, 2 f64, 3 i32),
//     u32 ndim, i64 shape[ndim], u64 nbytes, raw bytes.
// Records: "cfg" (i32 vc_config fields in declaration order + eps as the bit pattern of a float), "rope" (f64 [rows][cols][2]),
//          "x", "t", "geoada_context", "ctx.<i>", "seq_len" (i32) and "w.<state-dict key>" for every weight.
// out.bin: the raw bf16 bytes of the output [B, out_dim, T, H, W].
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../include/vcengine.h"

struct Rec {
    uint32_t kind = 0;
    std::vector<int64_t> shape;
    std::vector<char> bytes;
    void* dev = nullptr;
};

static bool read_exact(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n; }

#define CHECK_HIP(expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_)); return 2; } \
    } while (0)
#define CHECK_VC(h, expr)                                                                       \
    do {                                                                                        \
        int rc_ = (expr);                                                                       \
        if (rc_ != VC_OK) { fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, vc_last_error(h)); return 3; } \
    } while (0)

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s bundle.bin out.bin\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    uint32_t magic = 0, n = 0;
    if (!read_exact(f, &magic, 4) || !read_exact(f, &n, 4) || magic != 0x31424356u) { fprintf(stderr, "bad bundle\n"); return 1; }
    std::map<std::string, Rec> recs;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t nl = 0, nd = 0;
        Rec r;
        if (!read_exact(f, &nl, 4)) return 1;
        std::string name(nl, '\0');
        uint64_t nb = 0;
        if (!read_exact(f, &name[0], nl) || !read_exact(f, &r.kind, 4) || !read_exact(f, &nd, 4)) return 1;
        r.shape.resize(nd);
        if ((nd && !read_exact(f, r.shape.data(), 8 * nd)) || !read_exact(f, &nb, 8)) return 1;
        r.bytes.resize(nb);
        if (nb && !read_exact(f, r.bytes.data(), nb)) return 1;
        recs[name] = std::move(r);
    }
    fclose(f);
    auto upload = [&](Rec& r) -> int {
        CHECK_HIP(hipMalloc(&r.dev, r.bytes.size() ? r.bytes.size() : 4));
        CHECK_HIP(hipMemcpy(r.dev, r.bytes.data(), r.bytes.size(), hipMemcpyHostToDevice));
        return 0;
    };

    // ---- vc_create ----
    if (!recs.count("cfg") || recs["cfg"].bytes.size() < 12 * 4) { fprintf(stderr, "no cfg record\n"); return 1; }
    const int32_t* ci = (const int32_t*)recs["cfg"].bytes.data();
    vc_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.dim = ci[0]; cfg.ffn_dim = ci[1]; cfg.num_heads = ci[2]; cfg.num_layers = ci[3]; cfg.in_dim = ci[4]; cfg.out_dim = ci[5];
    cfg.geoada_in_dim = ci[6]; cfg.text_dim = ci[7]; cfg.text_len = ci[8]; cfg.freq_dim = ci[9];
    memcpy(&cfg.eps, &ci[10], 4);
    cfg.num_geoada_layers = 0;                               // default range(0, num_layers, 2) (VC.py:175)
    if (vc_abi_version() != VC_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    vc_engine* h = nullptr;
    { int rc = vc_create(&cfg, &h); if (rc != VC_OK) { fprintf(stderr, "vc_create -> %d: %s\n", rc, vc_last_error(nullptr)); return 3; } }

    // ---- weights: borrowed device pointers, addressed by the reference's state-dict keys ----
    int nw = 0;
    for (auto& kv : recs) {
        if (kv.first.rfind("w.", 0) != 0) continue;
        if (upload(kv.second)) return 2;
        CHECK_VC(h, vc_load_weight(h, kv.first.c_str() + 2, kv.second.dev, 0, (int)kv.second.shape.size(), kv.second.shape.data()));
        ++nw;
    }
    if (vc_missing_weights(h) != 0) { fprintf(stderr, "%d state-dict keys missing\n", vc_missing_weights(h)); return 3; }
    {
        Rec& r = recs["rope"];
        CHECK_VC(h, vc_set_rope_table(h, (const double*)r.bytes.data(), (int)r.shape[0], (int)r.shape[1]));
    }

    // ---- inputs ----
    Rec &x = recs["x"], &t = recs["t"], &geo = recs["geoada_context"];
    if (upload(x) || upload(t) || upload(geo)) return 2;
    const int B = (int)x.shape[0], T = (int)x.shape[2], H = (int)x.shape[3], W = (int)x.shape[4];
    std::vector<const void*> text(B);
    std::vector<int32_t> lens(B);
    for (int i = 0; i < B; ++i) {
        Rec& c = recs["ctx." + std::to_string(i)];
        if (upload(c)) return 2;
        text[i] = c.dev;
        lens[i] = (int32_t)c.shape[0];
    }
    const int32_t seq_len = *(const int32_t*)recs["seq_len"].bytes.data();
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    CHECK_VC(h, vc_prepare_video(h, geo.dev, text.data(), lens.data(), B, T, H, W, seq_len, stream));

    // ---- one forward (all blocks run: no TeaCache skip) ----
    const size_t out_bytes = (size_t)B * cfg.out_dim * T * H * W * 2;
    void* out = nullptr;
    CHECK_HIP(hipMalloc(&out, out_bytes));
    CHECK_VC(h, vc_forward(h, x.dev, (const float*)t.dev, out, 1.0f, VC_FWD_RUN_MAIN_BLOCKS, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    std::vector<char> host(out_bytes);
    CHECK_HIP(hipMemcpy(host.data(), out, out_bytes, hipMemcpyDeviceToHost));
    FILE* fo = fopen(argv[2], "wb");
    if (!fo || fwrite(host.data(), 1, out_bytes, fo) != out_bytes) { perror(argv[2]); return 1; }
    fclose(fo);
    printf("c_abi_denoise_step: %d weights, B=%d T=%d H=%d W=%d seq_len=%d, workspace %.1f MiB, wrote %zu bytes\n", nw, B, T, H, W,
           seq_len, vc_workspace_bytes(h) / 1048576.0, out_bytes);
    vc_destroy(h);
    return 0;
}
