// RCCL transport of the Ulysses exchange (include/vcengine.h: vc_sp_init_rccl).
//
// Stands in for the collectives of the un-vendored videox_fun.dist.usp_attn_forward / xFuserLongContextAttention that the
// reference binds at versecrafter/models/wan_transformer3d.py:901-921 (3 all-to-alls in, 1 out, per self-attention) and for
// get_sp_group().all_gather (wan_transformer3d_versecrafter.py:432-433).  The engine owns its communicators (one per block
// chain, see engine.hip) and enqueues the collective on the HIP stream of the chain that needs it: nothing crosses into
// Python on the step path, and the two chains never queue behind each other inside one communicator.
//
// librccl is bound at run time (dlopen; the instance the host process already loaded -- torch ships one -- is reused when
// there is one), so libvcengine.so has no link-time dependency on it and single-GPU users never load it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and enums only; every function is resolved through dlsym
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>

#include "vc_kernels.h"

namespace {

struct Api {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllToAll)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;   // RCCL extension
    std::string err;
};

Api g_api;
std::once_flag g_once;
thread_local std::string t_err;

void open_api() {
    Api& a = g_api;
    const char* forced = getenv("VC_RCCL_LIB");
    if (forced && *forced) {
        a.lib = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
    } else {
        // 1. the instance already mapped into this process (torch's librccl.so has soname librccl.so.1)
        a.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        // 2. the loader's search path, 3. the ROCm install
        if (!a.lib) a.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!a.lib) a.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    }
    if (!a.lib) {
        const char* e = dlerror();
        a.err = std::string("cannot load librccl.so.1: ") + (e ? e : "not found");
        return;
    }
    auto sym = [&](const char* n, bool required = true) -> void* {
        void* p = dlsym(a.lib, n);
        if (!p && required && a.err.empty()) a.err = std::string("librccl has no symbol ") + n;
        return p;
    };
    a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.CommCount = (decltype(a.CommCount))sym("ncclCommCount");
    a.CommUserRank = (decltype(a.CommUserRank))sym("ncclCommUserRank");
    a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
    a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
    a.Send = (decltype(a.Send))sym("ncclSend");
    a.Recv = (decltype(a.Recv))sym("ncclRecv");
    a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
    a.AllToAll = (decltype(a.AllToAll))sym("ncclAllToAll", false);
}

const Api* api() {
    std::call_once(g_once, open_api);
    if (!g_api.err.empty()) { t_err = g_api.err; return nullptr; }
    return &g_api;
}

int nfail(const Api* a, const char* what, ncclResult_t r) {
    t_err = std::string(what) + ": " + (a && a->GetErrorString ? a->GetErrorString(r) : "RCCL error") + " (" +
            std::to_string((int)r) + ")";
    return VC_E_HIP;
}

#define NCHK(a, expr)                                             \
    do {                                                          \
        ncclResult_t _r = (expr);                                 \
        if (_r != ncclSuccess) return nfail(a, #expr, _r);        \
    } while (0)

// the same inside an open ncclGroupStart: the group is closed before the error is returned -- an unclosed group would
// silently queue every later collective of this thread, a torch.distributed fallback's included
#define NCHK_IN_GROUP(a, expr)                                    \
    do {                                                          \
        ncclResult_t _r = (expr);                                 \
        if (_r != ncclSuccess) {                                  \
            (void)(a)->GroupEnd();                                \
            return nfail(a, #expr, _r);                           \
        }                                                         \
    } while (0)

}  // namespace

struct VcComm {
    ncclComm_t comm = nullptr;
    int world = 0, rank = 0;
    bool grouped_p2p = false;     // all-to-all as ncclGroupStart{Send/Recv per peer}GroupEnd instead of ncclAllToAll
};

const char* vc_comm_error() { return t_err.c_str(); }

int vc_comm_available() { return api() ? VC_OK : VC_E_UNSUPPORTED; }

int vc_comm_unique_id(void* out128) {
    const Api* a = api();
    if (!a) return VC_E_UNSUPPORTED;
    static_assert(sizeof(ncclUniqueId) == VC_RCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    NCHK(a, a->GetUniqueId(&id));
    memcpy(out128, &id, sizeof id);
    return VC_OK;
}

int vc_comm_create(VcComm** out, const void* id128, int world, int rank) {
    const Api* a = api();
    if (!a) return VC_E_UNSUPPORTED;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    VcComm* c = new VcComm();
    c->world = world; c->rank = rank;
    ncclResult_t r = a->CommInitRank(&c->comm, world, id, rank);     // blocks until every rank of the world has joined
    if (r != ncclSuccess) { delete c; return nfail(a, "ncclCommInitRank", r); }
    int n = 0, me = -1;
    if (a->CommCount(c->comm, &n) != ncclSuccess || a->CommUserRank(c->comm, &me) != ncclSuccess || n != world || me != rank) {
        t_err = "communicator reports " + std::to_string(n) + " ranks / rank " + std::to_string(me) + ", expected " +
                std::to_string(world) + " / " + std::to_string(rank);
        a->CommDestroy(c->comm);
        delete c;
        return VC_E_STATE;
    }
    const char* p2p = getenv("VC_SP_A2A");            // "p2p": grouped send/recv; default: ncclAllToAll when the library has it
    c->grouped_p2p = !a->AllToAll || (p2p && !strcmp(p2p, "p2p"));
    *out = c;
    return VC_OK;
}

int vc_comm_ranks(const VcComm* c) {
    const Api* a = api();
    int n = 0;
    if (!a || !c || a->CommCount(c->comm, &n) != ncclSuccess) return 0;
    return n;
}

void vc_comm_destroy(VcComm* c) {
    if (!c) return;
    const Api* a = api();
    if (a && c->comm) (void)a->CommDestroy(c->comm);
    delete c;
}

// peer r's slice is send[r * bytes_per_peer ...]; slice r of recv comes from rank r.  One message per peer: on the fully
// connected xGMI mesh every peer is one hop away on its own link, so the P-1 transfers of a rank run concurrently.
int vc_comm_all_to_all(VcComm* c, const void* send, void* recv, int64_t bytes_per_peer, hipStream_t s) {
    const Api* a = api();
    if (!a || !c) return VC_E_STATE;
    if (bytes_per_peer & 1) { t_err = "all_to_all: odd byte count"; return VC_E_INVALID; }
    const size_t cnt = (size_t)bytes_per_peer / 2;               // bf16 elements
    if (!c->grouped_p2p) {
        NCHK(a, a->AllToAll(send, recv, cnt, ncclBfloat16, c->comm, s));
        return VC_OK;
    }
    NCHK(a, a->GroupStart());
    for (int r = 0; r < c->world; ++r) {
        NCHK_IN_GROUP(a, a->Send((const char*)send + (int64_t)r * bytes_per_peer, cnt, ncclBfloat16, r, c->comm, s));
        NCHK_IN_GROUP(a, a->Recv((char*)recv + (int64_t)r * bytes_per_peer, cnt, ncclBfloat16, r, c->comm, s));
    }
    NCHK(a, a->GroupEnd());
    return VC_OK;
}

// `n` all-to-alls of equal shape on consecutive slabs (slab j = [P][bytes_per_peer] at offset j * P * bytes_per_peer of send and
// recv) as ONE RCCL group: one fused launch, every peer still one message per slab.
int vc_comm_all_to_all_n(VcComm* c, const void* send, void* recv, int64_t bytes_per_peer, int n, hipStream_t s) {
    const Api* a = api();
    if (!a || !c) return VC_E_STATE;
    if (n <= 1) return vc_comm_all_to_all(c, send, recv, bytes_per_peer, s);
    if (bytes_per_peer & 1) { t_err = "all_to_all: odd byte count"; return VC_E_INVALID; }
    const size_t cnt = (size_t)bytes_per_peer / 2;
    const int64_t slab = (int64_t)c->world * bytes_per_peer;
    NCHK(a, a->GroupStart());
    for (int j = 0; j < n; ++j) {
        const char* sj = (const char*)send + j * slab;
        char* rj = (char*)recv + j * slab;
        if (!c->grouped_p2p) {
            NCHK_IN_GROUP(a, a->AllToAll(sj, rj, cnt, ncclBfloat16, c->comm, s));
        } else {
            for (int r = 0; r < c->world; ++r) {
                NCHK_IN_GROUP(a, a->Send(sj + (int64_t)r * bytes_per_peer, cnt, ncclBfloat16, r, c->comm, s));
                NCHK_IN_GROUP(a, a->Recv(rj + (int64_t)r * bytes_per_peer, cnt, ncclBfloat16, r, c->comm, s));
            }
        }
    }
    NCHK(a, a->GroupEnd());
    return VC_OK;
}

// The two exchanges of the Ulysses x ring hybrid on the WORLD communicator (grouped point-to-point, one RCCL group each):
//   sub-group all-to-all: `n` slabs of [count][bytes_per_peer]; slice j of a slab goes to rank first + j, slice j of recv comes from it;
//   ring pass: `bytes` to rank dst, `bytes` from rank src (dst == src == own rank: a local copy).
int vc_comm_all_to_all_sub_n(VcComm* c, const void* send, void* recv, int64_t bytes_per_peer, int n, int first, int count, hipStream_t s) {
    const Api* a = api();
    if (!a || !c) return VC_E_STATE;
    if (first < 0 || count < 1 || first + count > c->world || n < 1) { t_err = "all_to_all_sub: bad rank range"; return VC_E_INVALID; }
    if (bytes_per_peer & 1) { t_err = "all_to_all_sub: odd byte count"; return VC_E_INVALID; }
    const size_t cnt = (size_t)bytes_per_peer / 2;
    const int64_t slab = (int64_t)count * bytes_per_peer;
    NCHK(a, a->GroupStart());
    for (int j = 0; j < n; ++j)
        for (int r = 0; r < count; ++r) {
            NCHK_IN_GROUP(a, a->Send((const char*)send + j * slab + (int64_t)r * bytes_per_peer, cnt, ncclBfloat16, first + r, c->comm, s));
            NCHK_IN_GROUP(a, a->Recv((char*)recv + j * slab + (int64_t)r * bytes_per_peer, cnt, ncclBfloat16, first + r, c->comm, s));
        }
    NCHK(a, a->GroupEnd());
    return VC_OK;
}

int vc_comm_sendrecv(VcComm* c, const void* send, int dst, void* recv, int src, int64_t bytes, hipStream_t s) {
    const Api* a = api();
    if (!a || !c) return VC_E_STATE;
    if (dst < 0 || dst >= c->world || src < 0 || src >= c->world || (bytes & 1)) { t_err = "sendrecv: bad argument"; return VC_E_INVALID; }
    NCHK(a, a->GroupStart());
    NCHK_IN_GROUP(a, a->Send(send, (size_t)bytes / 2, ncclBfloat16, dst, c->comm, s));
    NCHK_IN_GROUP(a, a->Recv(recv, (size_t)bytes / 2, ncclBfloat16, src, c->comm, s));
    NCHK(a, a->GroupEnd());
    return VC_OK;
}

int vc_comm_all_gather(VcComm* c, const void* send, void* recv, int64_t bytes, hipStream_t s) {
    const Api* a = api();
    if (!a || !c) return VC_E_STATE;
    if (bytes & 1) { t_err = "all_gather: odd byte count"; return VC_E_INVALID; }
    NCHK(a, a->AllGather(send, recv, (size_t)bytes / 2, ncclBfloat16, c->comm, s));
    return VC_OK;
}
