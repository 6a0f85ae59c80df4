"""Sequence parallelism (Ulysses) plumbing over torch.distributed (backend "nccl" = RCCL over xGMI).

Stands in for the parts of the un-vendored videox_fun.dist that the reference's hot path uses
(wan_transformer3d.py:30-32, 901-921; wan_transformer3d_versecrafter.py:269-270, 366-367, 432-433;
inference/versecrafter_inference.py:180): group set-up, rank / world size, the head-scatter
all-to-all around self-attention and the final all-gather.  The HIP engine packs / unpacks the exchange
buffers itself and runs the collectives on its own RCCL communicators (include/vcengine.h: vc_sp_init_rccl); the callback
transport (vc_sp_init) stays for the gloo tests.

Layout contract of one exchange (P ranks, B samples, Lloc = L/P local tokens, Nl = N/P local heads): one all-to-all per
(tensor, sample) slab, so that what arrives is already the plain [B][L][Nl][128] layout of the attention kernel --
    send  [3 (q,k,v)][B][P_dst][Lloc][Nl][128]  --3 B all_to_alls-->  recv [3][B][P_src][Lloc][Nl][128] = [3][B][L][Nl][128]
    attention over the full sequence (token t = src * Lloc + i) with Nl heads writes [B][L][Nl][128] =
    send  [B][P_dst (token owner)][Lloc][Nl][128] --B all_to_alls--> recv [B][P_src (head group)][Lloc][Nl][128]
(on RCCL the slabs of one exchange are one group = one fused launch; a rank still sends one message per peer and slab)
`pack_qkv` / `unpack_tokens` / `pack_out` / `unpack_heads` below restate that contract on torch tensors; the
multi-process CPU tests (gloo) drive them together with the byte-level collectives the engine uses.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib

_SP_GROUP = None
_BP_GROUP = None       # batch-parallel group (the samples of a CFG pair on different ranks); None = off
_RING_DEGREE = 1       # the ring degree the caller asked for (set_multi_gpus_devices); see choose_ring_degree
STALLED = False        # set when a communicator rendezvous did not return (SequenceParallelStall): a helper thread is still inside
                       # ncclCommInitRank with the engine handle -- nothing may destroy that handle any more, the process has to end
_RING_AS_GIVEN = False # True: a valid requested ring degree is used as it is (the reference's documented layout), never folded into Ulysses


def set_multi_gpus_devices(ulysses_degree: int, ring_degree: int, cfg_degree: int = 1, ring_as_given: bool = False):
    """CLI.py:180.  One process per GPU (torchrun env); returns this rank's device.  The reference's
    ulysses x ring hybrid is run as pure Ulysses of degree ulysses*ring when the model's head count divides by it (14B: 40 heads),
    else as the hybrid with a ring of `ring_degree` (choose_ring_degree; the engine's vc_sp_set_ring).

    cfg_degree (this build; the reference has no such split): the samples of one forward's batch -- the [uncond, cond] pair of
    classifier-free guidance, PIPE.py:878-887 -- are independent units, so `cfg_degree` ranks can take one sample each with no
    data-path collective at all; only the noise prediction (4 MB at cfg-3) is all-gathered after the forward.  World size =
    cfg_degree * ulysses_degree * ring_degree; rank r works on sample r // S inside the Ulysses group of its S = ulysses * ring
    neighbours [r - r % S, r - r % S + S).

    ring_as_given (this build; CLI `--sp_layout as_given`): run exactly `ulysses_degree x ring_degree` -- the reference's documented
    launch line (`inference.sh:62-71`: 2 x 4 / 4 x 2) -- even where pure Ulysses of the product would fit the head count, so that the two
    layouts can be compared on hardware with one flag."""
    global _RING_DEGREE, _RING_AS_GIVEN
    degree = int(ulysses_degree) * int(ring_degree)
    _RING_DEGREE = int(ring_degree)
    _RING_AS_GIVEN = bool(ring_as_given)
    cfg_degree = int(cfg_degree)
    world = degree * cfg_degree
    if world > 1:
        ensure_ipc_env()
        if not dist.is_initialized():
            # host-side gloo for rendezvous / object broadcast (the ncclUniqueIds of the engine's communicators), torch's
            # "nccl" (= RCCL) registered for device tensors; the engine's exchanges never go through torch
            # VC_DIST_BACKEND=gloo: a launch rehearsal with several ranks on one device (RCCL refuses two ranks per GPU; gloo groups
            # stage the exchange buffers through host memory, _host_bounce) -- tests only
            dist.init_process_group(os.environ.get("VC_DIST_BACKEND") or
                                    ("cpu:gloo,cuda:nccl" if torch.cuda.is_available() else "gloo"))
        if dist.get_world_size() != world:
            raise ValueError(f"cfg_degree*ulysses_degree*ring_degree = {world} but world size is {dist.get_world_size()}")
        if int(ring_degree) > 1 and dist.get_rank() == 0 and not ring_as_given:
            print(f"[versecrafter_amd] ulysses_degree={ulysses_degree} x ring_degree={ring_degree}: run as pure Ulysses of degree {degree} when the "
                  "model's head count divides by it, else as the Ulysses x ring hybrid (dist.choose_ring_degree)")
        make_groups(degree, cfg_degree)
    local = int(os.environ.get("LOCAL_RANK", 0))
    if torch.cuda.is_available():
        if os.environ.get("VC_DIST_BACKEND") == "gloo":
            local %= torch.cuda.device_count()                   # rehearsal: ranks share devices
        torch.cuda.set_device(local)
        return torch.device("cuda", local)
    return torch.device("cpu")


def ensure_ipc_env():
    """HSA_ENABLE_IPC_MODE_LEGACY=0 before the first HIP call of a multi-process GPU job.  The hosts of this pool support only
    dmabuf IPC: with the legacy mode RCCL's intra-node transport (and any CUDA-tensor sharing across processes) fails with
    `hipIpcGetMemHandle: invalid argument` (the environment notes of this build; the variable is exported on the image already,
    this keeps it for launchers that scrub the environment).  A value the user has set is left alone."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def make_groups(sp_degree: int, cfg_degree: int = 1):
    """The process groups of set_multi_gpus_devices on an initialised world of cfg_degree * sp_degree ranks (callers that pick
    their own devices: bench.py's one-GPU rehearsal).  Returns (sp_group or None, bp_group or None)."""
    global _SP_GROUP, _BP_GROUP
    world = dist.get_world_size()
    if world != sp_degree * cfg_degree:
        raise ValueError(f"cfg_degree*sp_degree = {sp_degree * cfg_degree} but world size is {world}")
    _SP_GROUP = _BP_GROUP = None
    if cfg_degree == 1:
        _SP_GROUP = dist.group.WORLD
        return _SP_GROUP, None
    me = dist.get_rank()
    for c in range(cfg_degree):                   # every rank creates every group (new_group is collective)
        ranks = list(range(c * sp_degree, (c + 1) * sp_degree))
        g = dist.new_group(ranks=ranks) if sp_degree > 1 else None
        if me in ranks:
            _SP_GROUP = g
    for s_ in range(sp_degree):
        ranks = list(range(s_, world, sp_degree))
        g = dist.new_group(ranks=ranks)
        if me in ranks:
            _BP_GROUP = g
    return _SP_GROUP, _BP_GROUP


def get_ring_degree() -> int:
    return _RING_DEGREE


def ring_as_given() -> bool:
    return _RING_AS_GIVEN


def choose_ring_degree(world: int, num_heads: int, requested: int = 1, as_given: bool = None) -> int:
    """Ring degree R of a sequence-parallel world (Ulysses degree U = world / R must divide num_heads).
    VC_SP_RING forces one (tests).  Otherwise: 1 -- pure Ulysses, two all-to-alls per attention and no partial-output merge -- whenever
    the head count allows it (Wan2.1-14B: 40 heads, any world in {1, 2, 4, 5, 8}; this is how the reference's default 4 x 2 has always
    been run here); else the requested degree if it is valid, else the smallest valid one (Wan2.1-1.3B, 12 heads, on 8 ranks: 2)."""
    forced = os.environ.get("VC_SP_RING")
    if forced:
        r = int(forced)
        if r < 1 or world % r or num_heads % (world // r):
            raise ValueError(f"VC_SP_RING={r}: world {world} / ring must divide num_heads {num_heads}")
        return r
    if as_given is None:
        as_given = _RING_AS_GIVEN
    if as_given and requested >= 1:         # the caller's U x R exactly (set_multi_gpus_devices(ring_as_given=True))
        if world % requested or num_heads % (world // requested) or requested > 8:
            raise ValueError(f"ring degree {requested}: world {world} / ring must divide num_heads {num_heads} (and the ring be <= 8)")
        return requested
    if num_heads % world == 0:
        return 1
    valid = [r for r in range(2, min(world, 8) + 1) if world % r == 0 and num_heads % (world // r) == 0]
    if not valid:
        raise ValueError(f"no Ulysses x ring split of {world} ranks fits {num_heads} heads")
    return requested if requested in valid else valid[0]


def use_groups(sp_group, bp_group):
    """Select a pair of groups made earlier by make_groups (a program that measures several layouts on one world)."""
    global _SP_GROUP, _BP_GROUP
    _SP_GROUP, _BP_GROUP = sp_group, bp_group


def get_sp_group():
    return _SP_GROUP


def get_bp_group():
    return _BP_GROUP


class BatchParallel:
    """The samples of one forward's batch on different ranks (see set_multi_gpus_devices: cfg_degree).  Nothing of the forward
    itself is exchanged; `gather` puts the per-rank outputs back together, `broadcast` hands one rank's output to the group
    (steps on which only the conditional sample is computed, cfg_skip).  gloo groups (tests) stage device tensors through host."""

    def __init__(self, group):
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def gather(self, x: torch.Tensor) -> torch.Tensor:
        """x [1, ...] of this rank's sample -> [world_size, ...] in rank order."""
        x = x.contiguous()
        if _host_bounce(x, self.group):
            h = x.cpu()
            parts = [torch.empty_like(h) for _ in range(self.world_size)]
            dist.all_gather(parts, h, group=self.group)
            return torch.cat(parts, dim=0).to(x.device)
        out = torch.empty((self.world_size * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x, group=self.group)
        return out

    def observed_ranks(self, device) -> dict:
        """Count the ranks of this group by running a collective: every rank contributes 1 to an all-reduce on the group's
        DEVICE backend when it has one ("nccl" = RCCL; the sum is read back), else on the host backend.  What the transport
        saw, not what the launcher computed."""
        on_device = torch.device(device).type == "cuda" and not dist.get_backend(self.group) == "gloo"
        one = torch.ones(1, dtype=torch.float32, device=device if on_device else "cpu")
        dist.all_reduce(one, group=self.group)
        return {"ranks": int(round(float(one.item()))), "backend": "nccl" if on_device else "gloo"}

    def broadcast(self, x: torch.Tensor, src: int) -> torch.Tensor:
        """x from group rank `src` to everyone (in place on the other ranks)."""
        g_src = dist.get_global_rank(self.group, src)
        if _host_bounce(x, self.group):
            h = x.cpu()
            dist.broadcast(h, src=g_src, group=self.group)
            if self.rank != src:
                x.copy_(h)
            return x
        dist.broadcast(x, src=g_src, group=self.group)
        return x


def get_sequence_parallel_world_size():
    return 1 if _SP_GROUP is None else dist.get_world_size(_SP_GROUP)


def get_sequence_parallel_rank():
    return 0 if _SP_GROUP is None else dist.get_rank(_SP_GROUP)


# ---- byte-level collectives (device-agnostic torch plumbing) --------------------------------------
def _host_bounce(t: torch.Tensor, group) -> bool:
    """gloo has no device all-to-all / all-gather: device buffers are staged through host memory.  This is the functional
    transport of the multi-process tests on one GPU; the product transport is RCCL ("nccl"), which never takes it."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_to_all_bytes(send: torch.Tensor, recv: torch.Tensor, group=None):
    """send / recv: flat uint8 tensors of P equal slices; slice r of `send` goes to rank r."""
    if _host_bounce(send, group):
        h_send = send.cpu()                      # waits for the producer kernels on the current stream
        h_recv = torch.empty_like(h_send)
        dist.all_to_all_single(h_recv, h_send, group=group)
        recv.copy_(h_recv)
        return
    dist.all_to_all_single(recv, send, group=group)


def all_gather_bytes(send: torch.Tensor, recv: torch.Tensor, group=None):
    """recv (P * len(send)) = concatenation over ranks of `send`."""
    if _host_bounce(send, group):
        h_send = send.cpu()
        h_recv = torch.empty(recv.numel(), dtype=recv.dtype)
        dist.all_gather_into_tensor(h_recv, h_send, group=group)
        recv.copy_(h_recv)
        return
    dist.all_gather_into_tensor(recv, send, group=group)


# ---- the exchange-buffer layout contract, on tensors ----------------------------------------------
def pack_qkv(qkv: torch.Tensor, P: int) -> torch.Tensor:
    """qkv [B, Lloc, 3, N, D] (local tokens, all heads) -> send [3, B, P_dst, Lloc, N/P, D]."""
    B, Lloc, three, N, D = qkv.shape
    return qkv.view(B, Lloc, 3, P, N // P, D).permute(2, 0, 3, 1, 4, 5).contiguous()


def unpack_tokens(recv: torch.Tensor):
    """recv [3, B, P_src, Lloc, Nl, D] -> q, k, v each [B, P*Lloc, Nl, D] (token t = src*Lloc + i): a view, no data movement."""
    three, B, P, Lloc, Nl, D = recv.shape
    full = recv.reshape(3, B, P * Lloc, Nl, D)
    return full[0], full[1], full[2]


def pack_out(o: torch.Tensor, P: int) -> torch.Tensor:
    """o [B, P*Lloc, Nl, D] -> send [B, P_dst, Lloc, Nl, D]: a view."""
    B, L, Nl, D = o.shape
    return o.contiguous().view(B, P, L // P, Nl, D)


def unpack_heads(recv: torch.Tensor) -> torch.Tensor:
    """recv [B, P_src, Lloc, Nl, D] -> [B, Lloc, P*Nl, D] (head = src*Nl + hl)."""
    B, P, Lloc, Nl, D = recv.shape
    return recv.permute(0, 2, 1, 3, 4).reshape(B, Lloc, P * Nl, D)


def all_to_all_slabs(send: torch.Tensor, recv: torch.Tensor, nslab: int, group=None):
    """`nslab` all-to-alls on the leading slabs of send / recv (each slab = P equal pieces)."""
    s8, r8 = send.view(torch.uint8).flatten(), recv.view(torch.uint8).flatten()
    n = s8.numel() // nslab
    for j in range(nslab):
        all_to_all_bytes(s8[j * n:(j + 1) * n], r8[j * n:(j + 1) * n], group)


def ulysses_attention(q, k, v, attn_fn, group=None):
    """Reference semantics of usp_attn_forward's exchange (third-party) on local
    [B, Lloc, N, D] tensors: returns the local [B, Lloc, N, D] attention output."""
    P = dist.get_world_size(group)
    B, Lloc, N, D = q.shape
    send = pack_qkv(torch.stack([q, k, v], dim=2), P)
    recv = torch.empty_like(send)
    all_to_all_slabs(send, recv, 3 * B, group)
    qf, kf, vf = unpack_tokens(recv)
    o = attn_fn(qf, kf, vf)
    send2 = pack_out(o.contiguous(), P)
    recv2 = torch.empty_like(send2)
    all_to_all_slabs(send2, recv2, B, group)
    return unpack_heads(recv2)


def hybrid_attention(q, k, v, attn_lse_fn, ring: int, group=None, k_len=None):
    """Reference semantics of the Ulysses x ring hybrid (third-party xFuserLongContextAttention) on local
    [B, Lloc, N, D] tensors of rank g * U + u in a world of U * ring ranks: all-to-all inside the Ulysses group (heads scattered, the
    group's tokens gathered), K|V blocks passed round the ring of the ranks that share u, the partial outputs merged by their
    log-sum-exps, inverse all-to-all.  attn_lse_fn(q, k, v, k_len) -> (out [B, Lq, Nu, D], lse [B, Nu, Lq], natural log; k_len valid keys).
    k_len: valid tokens of the whole sequence (None = all).  The CPU tests drive this with the oracle's attention."""
    P, me = dist.get_world_size(group), dist.get_rank(group)
    U = P // ring
    g, u = me // U, me % U
    B, Lloc, N, D = q.shape
    Nu, Lg = N // U, U * Lloc
    peers = [dist.get_global_rank(group, g * U + j) if group is not None else g * U + j for j in range(U)]

    def a2a_sub(x):                                    # x [U_dst, ...] -> [U_src, ...]
        out = torch.empty_like(x)
        ops_ = []
        for j in range(U):
            if g * U + j == me:
                out[j].copy_(x[j])
            else:
                ops_ += [dist.P2POp(dist.isend, x[j].contiguous(), peers[j], group), dist.P2POp(dist.irecv, out[j], peers[j], group)]
        for w in (dist.batch_isend_irecv(ops_) if ops_ else []):
            w.wait()
        return out
    send = torch.stack([q, k, v], 0).view(3, B, Lloc, U, Nu, D).permute(3, 0, 1, 2, 4, 5).contiguous()      # [U_dst, 3, B, Lloc, Nu, D]
    recv = a2a_sub(send)                                                                                   # [U_src, 3, B, Lloc, Nu, D]
    full = recv.permute(1, 2, 0, 3, 4, 5).reshape(3, B, Lg, Nu, D)                                         # token = src * Lloc + i
    qg, kv = full[0], full[1:].contiguous()
    outs, lses = [], []
    nxt = dist.get_global_rank(group, ((g + 1) % ring) * U + u) if group is not None else ((g + 1) % ring) * U + u
    prv = dist.get_global_rank(group, ((g - 1) % ring) * U + u) if group is not None else ((g - 1) % ring) * U + u
    for st in range(ring):
        gb = (g - st) % ring
        kl = Lg if k_len is None else max(0, min(Lg, k_len - gb * Lg))
        if kl > 0:
            o_, l_ = attn_lse_fn(qg, kv[0], kv[1], kl)
        else:
            o_, l_ = torch.zeros_like(qg), torch.full((B, Nu, Lg), float("-inf"))
        outs.append(o_)
        lses.append(l_)
        if st + 1 < ring:
            got = torch.empty_like(kv)
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, kv, nxt, group), dist.P2POp(dist.irecv, got, prv, group)]):
                w.wait()
            kv = got
    lse = torch.stack(lses, 0)                                                   # [R, B, Nu, Lg]
    w = torch.softmax(lse, dim=0)                                                # exp(lse_r - logsumexp)
    o = sum(w[r].permute(0, 2, 1).unsqueeze(-1) * outs[r] for r in range(ring))  # [B, Lg, Nu, D]
    back = a2a_sub(o.view(B, U, Lloc, Nu, D).permute(1, 0, 2, 3, 4).contiguous())             # [U_src (head group), B, Lloc, Nu, D]
    return back.permute(1, 2, 0, 3, 4).reshape(B, Lloc, N, D)


class _DevBuf:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can alias it (no copy)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def alias_device_bytes(ptr: int, nbytes: int, device) -> torch.Tensor:
    return torch.as_tensor(_DevBuf(ptr, nbytes), device=device)


def _group_has_rccl(group) -> bool:
    """True when the group's device backend is torch's "nccl" (= RCCL on ROCm)."""
    try:
        return "nccl" in str(dist.get_backend_config(group))
    except Exception:
        return dist.get_backend(group) == "nccl"


class SequenceParallelStall(RuntimeError):
    """The blocking rendezvous of the engine's RCCL communicators did not return within its deadline.  The helper thread is still inside
    ncclCommInitRank with the engine handle: no clean-up may touch that handle (the model's __del__ skips vc_destroy once `STALLED` is
    set); a program that catches this should print it and leave with os._exit(non-zero) -- never exec -- so that torchrun / the bench's
    supervisor starts fresh ranks (inference/versecrafter_inference.py does)."""


class SequenceParallel:
    """Wires the engine's Ulysses exchange to a transport (include/vcengine.h):

    "rccl"  (product, default when the group's device backend is nccl): the ENGINE owns two RCCL communicators, one per
            block chain, and enqueues ncclAllToAll / ncclAllGather on the chain's HIP stream itself (vc_sp_init_rccl);
            this class only ships the two ncclUniqueIds from rank 0 to the other ranks over the torch group.
    "torch" (tests on gloo with host-staged buffers; VC_SP_TRANSPORT=torch forces it on nccl): the engine calls back into
            this class with raw pointers into its workspace, which are aliased as uint8 torch tensors and exchanged with
            torch.distributed on the stream the engine passes in.  On nccl each chain gets its own process group --
            ProcessGroupNCCL runs all collectives of one group on one internal stream in issue order, so a shared group
            would make the main chain's q|k|v exchange queue behind the adapter chain's."""

    def __init__(self, group=None, transport=None, force_exchange=False, ring_degree=1):
        self.ring_degree = int(ring_degree)
        self.group = group if group is not None else get_sp_group()
        if self.group is None and dist.is_initialized() and _BP_GROUP is None:
            self.group = dist.group.WORLD
        self.world_size = 1 if self.group is None else dist.get_world_size(self.group)
        self.rank = 0 if self.group is None else dist.get_rank(self.group)
        self.error = None
        self.force_exchange = bool(force_exchange)
        if transport is None:
            transport = os.environ.get("VC_SP_TRANSPORT") or None
        if transport is None:
            rccl = torch.cuda.is_available() and (self.group is None or _group_has_rccl(self.group))
            transport = "rccl" if rccl and (self.world_size > 1 or self.force_exchange) else "torch"
        if transport not in ("rccl", "torch"):
            raise ValueError(f"unknown sequence-parallel transport {transport!r}")
        self.transport = transport
        self._alias = {}
        self._lane_of_stream = {}
        self._lane_groups = [self.group]
        if transport == "torch" and self.world_size > 1 and _group_has_rccl(self.group):
            ranks = [dist.get_global_rank(self.group, i) for i in range(self.world_size)]
            self._lane_groups.append(dist.new_group(ranks=ranks, backend="nccl"))     # collective: every rank gets here
        self.c_all_to_all = _lib.ALL_TO_ALL_FN(self._a2a)
        self.c_all_gather = _lib.ALL_GATHER_FN(self._ag)
        self.c_all_to_all_sub = _lib.ALL_TO_ALL_SUB_FN(self._a2a_sub)
        self.c_sendrecv = _lib.SENDRECV_FN(self._sendrecv)
        if self.ring_degree < 1 or self.world_size % self.ring_degree:
            raise ValueError(f"ring degree {self.ring_degree} must divide the sequence-parallel world of {self.world_size} ranks")

    def _set_ring(self, lib, handle):
        """Ulysses x ring hybrid (vc_sp_set_ring): on the engine-owned RCCL transport the engine does both exchanges itself
        (grouped point-to-point on the world communicator); on the torch transport through the two callbacks below."""
        if self.ring_degree > 1:
            if self.transport == "rccl":
                _lib.check(lib.vc_sp_set_ring(handle, self.ring_degree, _lib.ALL_TO_ALL_SUB_FN(), _lib.SENDRECV_FN()), handle)
            else:
                _lib.check(lib.vc_sp_set_ring(handle, self.ring_degree, self.c_all_to_all_sub, self.c_sendrecv), handle)

    def _agree(self, failed: bool) -> bool:
        """True when ANY rank of the group reports a failure (one host-side all-reduce; identity at world 1)."""
        if self.world_size <= 1:
            return bool(failed)
        host_side = "gloo" in str(dist.get_backend_config(self.group))          # keep torch's own RCCL communicator unborn
        flag = torch.tensor([1 if failed else 0], device="cpu" if host_side else "cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
        return bool(flag.item())

    def attach(self, lib, handle):
        """vc_sp_init / vc_sp_init_rccl on one engine handle (called by the model when the engine is (re)configured).

        The ranks AGREE before every step that can fail on one side only, so that no rank is ever alone inside a collective:
          1. every rank binds librccl (vc_rccl_available) and rank 0 also creates the two ncclUniqueIds -> all-reduce of a
             failure flag.  Any failure: every rank takes the torch transport (still RCCL, through torch.distributed's "nccl"
             groups) or, on a group without an RCCL backend, every rank raises -- together.
          2. the ids are broadcast and every rank calls vc_sp_init_rccl (ncclCommInitRank: a blocking rendezvous).  It runs on
             a helper thread with a deadline (VC_SP_INIT_TIMEOUT seconds, default 120): a rank whose rendezvous does not
             return raises SequenceParallelStall -- the call cannot be cancelled, so the process must end; bench.py's
             supervisor then starts a FRESH set of ranks with VC_SP_TRANSPORT=torch (never a re-exec of a process that has
             touched the GPU).
          3. a second all-reduce of a failure flag (ncclCommInitRank returned an error somewhere): the same fallback."""
        self.error = None
        if self.transport == "rccl":
            n = _lib.VC_RCCL_UNIQUE_ID_BYTES
            ids = C.create_string_buffer(2 * n)
            err = None
            try:                                                        # step 1: nothing here blocks or involves a peer
                _lib.check(lib.vc_rccl_available())
                if self.rank == 0:
                    for i in range(2):
                        _lib.check(lib.vc_rccl_unique_id(C.byref(ids, i * n), n))
            except Exception as e:                                      # library missing, symbol missing, wrong VC_RCCL_LIB ...
                err = e
            failed = self._agree(err is not None)
            if not failed:
                try:                                                    # step 2
                    if self.world_size > 1:
                        box = [ids.raw if self.rank == 0 else None]
                        host_side = "gloo" in str(dist.get_backend_config(self.group))
                        dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0), group=self.group,
                                                   device=torch.device("cpu") if host_side else None)
                        ids = C.create_string_buffer(box[0], 2 * n)
                    flags = _lib.VC_SP_FORCE_EXCHANGE if self.force_exchange else 0
                    self._init_rccl_with_deadline(lib, handle, ids, flags)
                except SequenceParallelStall:
                    raise
                except Exception as e:
                    err = e
                failed = self._agree(err is not None)                   # step 3
            if not failed:
                self._set_ring(lib, handle)
                return
            if self.world_size == 1 or not _group_has_rccl(self.group):
                raise err if err is not None else RuntimeError("engine-owned RCCL transport failed on another rank")
            import sys
            print(f"[versecrafter_amd] rank {self.rank}: engine-owned RCCL communicators unavailable "
                  f"({err if err is not None else 'failure on another rank'}); falling back to torch.distributed's RCCL "
                  "process groups", file=sys.stderr)
            self.transport = "torch"
            if len(self._lane_groups) == 1:
                ranks = [dist.get_global_rank(self.group, i) for i in range(self.world_size)]
                self._lane_groups.append(dist.new_group(ranks=ranks, backend="nccl"))
        _lib.check(lib.vc_sp_init(handle, self.world_size, self.rank, self.c_all_to_all, self.c_all_gather, None), handle)
        self._set_ring(lib, handle)

    def _init_rccl_with_deadline(self, lib, handle, ids, flags):
        """vc_sp_init_rccl on a helper thread; SequenceParallelStall when it has not returned within the deadline."""
        import threading
        limit = float(os.environ.get("VC_SP_INIT_TIMEOUT", "120"))
        box = {}

        def run():
            try:
                box["rc"] = lib.vc_sp_init_rccl(handle, self.world_size, self.rank, ids, 2, flags)
            except BaseException as e:          # noqa: BLE001 -- handed to the caller's thread
                box["exc"] = e

        dev = torch.cuda.current_device() if torch.cuda.is_available() else None
        th = threading.Thread(target=lambda: (torch.cuda.set_device(dev) if dev is not None else None, run()), daemon=True,
                              name="vc_sp_init_rccl")
        th.start()
        th.join(limit if limit > 0 else None)
        if th.is_alive():
            global STALLED
            STALLED = True
            raise SequenceParallelStall(
                f"rank {self.rank}: ncclCommInitRank of the engine's communicators has not returned after {limit:.0f} s "
                f"(world {self.world_size}).  The call cannot be cancelled: end this process and start fresh ranks with "
                "VC_SP_TRANSPORT=torch (torch.distributed's own RCCL groups) or VC_SP_A2A=p2p; bench.py does that by itself.")
        if "exc" in box:
            raise box["exc"]
        _lib.check(box["rc"], handle)

    def comm_ranks(self, lib, handle) -> int:
        """World size the engine's RCCL communicator reports (ncclCommCount); 0 on the callback transport."""
        return int(lib.vc_sp_comm_ranks(handle))

    def observed_ranks(self, handle, device) -> int:
        """Ranks counted BY the transport: ncclCommCount of the engine's communicator ("rccl"); on the torch transport the sum
        of an all-reduce of ones over the lane group the exchanges run on (RCCL when the group has it, else gloo)."""
        if self.transport == "rccl":
            return self.comm_ranks(_lib.load(), handle)
        if self.world_size == 1:
            return 1
        grp = self._lane_groups[-1]
        on_device = torch.device(device).type == "cuda" and dist.get_backend(grp) != "gloo"
        one = torch.ones(1, dtype=torch.float32, device=device if on_device else "cpu")
        dist.all_reduce(one, group=grp)
        return int(round(float(one.item())))

    def probe(self, lib, handle, device) -> dict:
        """One small all-to-all on each chain's communicator and one all-gather through the ENGINE's own entry points
        (vc_sp_all_to_all / vc_sp_all_gather: the calls the step path makes), with the payload checked: slice r of what rank
        `me` receives must be what rank r addressed to `me`.  Raises on a wrong byte; a transport that hangs here hangs inside
        the supervised bring-up window of bench.py, not in the timed region."""
        P, me, n = self.world_size, self.rank, 128                         # n bf16 elements per peer
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        with torch.cuda.device(device):
            for chain in (0, 1):                      # two slabs per exchange: the grouped form the step path uses
                base = torch.arange(P, device=device, dtype=torch.float32)
                send = torch.cat([(base + me * P + 64 * chain + 32 * j).repeat_interleave(n) for j in (0, 1)]).bfloat16()
                recv = torch.full_like(send, -1.0)
                _lib.check(lib.vc_sp_all_to_all_n(handle, chain, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), 2 * n, 2,
                                                  stream), handle)
                want = torch.cat([(base * P + me + 64 * chain + 32 * j).repeat_interleave(n) for j in (0, 1)]).bfloat16()
                if not torch.equal(recv, want):
                    raise RuntimeError(f"sequence-parallel probe: all-to-all on chain {chain} delivered wrong data on rank {me}")
            send = torch.full((n,), float(me), device=device).bfloat16()
            recv = torch.full((P * n,), -1.0, device=device).bfloat16()
            _lib.check(lib.vc_sp_all_gather(handle, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), 2 * n, stream), handle)
            want = torch.arange(P, device=device, dtype=torch.float32).repeat_interleave(n).bfloat16()
            if not torch.equal(recv, want):
                raise RuntimeError(f"sequence-parallel probe: all-gather delivered wrong data on rank {me}")
            if self.ring_degree > 1:                   # the hybrid's two exchanges: sub-group all-to-all, ring pass
                U = P // self.ring_degree
                g_, u_ = me // U, me % U
                for chain in (0, 1):
                    base = torch.arange(U, device=device, dtype=torch.float32)
                    send = (base + u_ * U + 16 * g_ + 64 * chain).repeat_interleave(n).bfloat16()
                    recv = torch.full_like(send, -1.0)
                    _lib.check(lib.vc_sp_all_to_all_sub(handle, chain, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), 2 * n, 1, g_ * U,
                                                        U, stream), handle)
                    want = (base * U + u_ + 16 * g_ + 64 * chain).repeat_interleave(n).bfloat16()
                    if not torch.equal(recv, want):
                        raise RuntimeError(f"sequence-parallel probe: sub-group all-to-all on chain {chain} delivered wrong data on rank {me}")
                    R_ = self.ring_degree
                    send = torch.full((n,), float(me + 100 * chain), device=device).bfloat16()
                    recv = torch.full_like(send, -1.0)
                    src = ((g_ - 1) % R_) * U + u_
                    _lib.check(lib.vc_sp_sendrecv(handle, chain, C.c_void_p(send.data_ptr()), ((g_ + 1) % R_) * U + u_,
                                                  C.c_void_p(recv.data_ptr()), src, 2 * n, stream), handle)
                    if not torch.equal(recv, torch.full((n,), float(src + 100 * chain), device=device).bfloat16()):
                        raise RuntimeError(f"sequence-parallel probe: ring pass on chain {chain} delivered wrong data on rank {me}")
        if self.error is not None:
            raise RuntimeError("sequence-parallel probe: a collective failed") from self.error
        return {"ranks": self.observed_ranks(handle, device), "transport": self.transport, "ring_degree": self.ring_degree}

    def _buf(self, ptr, nbytes):
        key = (ptr, nbytes)
        t = self._alias.get(key)
        if t is None:
            t = alias_device_bytes(ptr, nbytes, torch.device("cuda", torch.cuda.current_device()))
            self._alias[key] = t
        return t

    def _group_for(self, stream):
        """The engine calls back with one stream per lane (numbered in order of first appearance: deterministic, the same
        program runs on every rank) -- two under the lane schedules, three under the sample pipeline (the caller's stream for
        the batched first block and the final gather, then one exchange stream per sample).  Lane k uses process group
        k mod 2, which keeps the two samples' exchanges on different groups in every schedule."""
        lane = self._lane_of_stream.setdefault(stream, len(self._lane_of_stream))
        return self._lane_groups[lane % len(self._lane_groups)]

    @staticmethod
    def _on(stream):
        """Run torch ops on the HIP stream the engine enqueued the surrounding kernels on (its main chain uses the
        caller's stream, the GeoAdapter chain an engine-owned one)."""
        if not stream:
            return torch.cuda.stream(torch.cuda.default_stream())
        return torch.cuda.stream(torch.cuda.ExternalStream(stream))

    def _a2a(self, ctx, send, recv, bytes_per_peer, stream):
        try:
            n = bytes_per_peer * self.world_size
            with self._on(stream):
                all_to_all_bytes(self._buf(send, n), self._buf(recv, n), self._group_for(stream))
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            self.error = e
            return -1

    def _ag(self, ctx, send, recv, nbytes, stream):
        try:
            with self._on(stream):
                all_gather_bytes(self._buf(send, nbytes), self._buf(recv, nbytes * self.world_size), self._group_for(stream))
            return 0
        except Exception as e:
            self.error = e
            return -1

    def _p2p(self, ops_spec, stream):
        """[(send tensor or None, dst, recv tensor or None, src)] as one batch of point-to-point operations on the lane's group
        (RCCL when it has it; gloo stages device buffers through host memory).  Ranks are group ranks."""
        grp = self._group_for(stream)
        bounce = dist.get_backend(grp) == "gloo"
        reqs, staged = [], []
        with self._on(stream):
            for snd, dst, rcv, src in ops_spec:
                if snd is not None:
                    t = snd.cpu() if bounce else snd
                    reqs.append(dist.P2POp(dist.isend, t, dist.get_global_rank(grp, dst), grp))
                if rcv is not None:
                    t = torch.empty(rcv.numel(), dtype=rcv.dtype) if bounce else rcv
                    if bounce:
                        staged.append((rcv, t))
                    reqs.append(dist.P2POp(dist.irecv, t, dist.get_global_rank(grp, src), grp))
            for w in (dist.batch_isend_irecv(reqs) if reqs else []):
                w.wait()
            for rcv, t in staged:
                rcv.copy_(t)

    def _a2a_sub(self, ctx, send, recv, bytes_per_peer, first, count, stream):
        try:
            n = bytes_per_peer * count
            s8, r8 = self._buf(send, n), self._buf(recv, n)
            spec = []
            with self._on(stream):
                for j in range(count):
                    sl = slice(j * bytes_per_peer, (j + 1) * bytes_per_peer)
                    if first + j == self.rank:
                        r8[sl].copy_(s8[sl])
                    else:
                        spec.append((s8[sl], first + j, r8[sl], first + j))
            self._p2p(spec, stream)
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            self.error = e
            return -1

    def _sendrecv(self, ctx, send, dst, recv, src, nbytes, stream):
        try:
            s8, r8 = self._buf(send, nbytes), self._buf(recv, nbytes)
            if dst == self.rank and src == self.rank:
                with self._on(stream):
                    r8.copy_(s8)
            else:
                self._p2p([(s8, dst, r8, src)], stream)
            return 0
        except Exception as e:
            self.error = e
            return -1

    def all_gather_dim1(self, x: torch.Tensor, dim: int = 1) -> torch.Tensor:
        """get_sp_group().all_gather(x, dim=1) of the reference (VC.py:432-433)."""
        if self.world_size == 1:
            return x
        parts = [torch.empty_like(x) for _ in range(self.world_size)]
        dist.all_gather(parts, x.contiguous(), group=self.group)
        return torch.cat(parts, dim=dim)
