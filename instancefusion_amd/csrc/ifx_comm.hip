// ifx_comm.hip -- the collectives of a spatially sharded map, inside the library's fixed schedule.
//
// A handle created with ifx_config::n_ranks = G stores 1 / G of the map; between the phases of a frame the ranks reduce the buffers
// ifx_owner_exchange(phase) names.  Round 2 left that to the caller (eight returns to Python per frame, torch.distributed in between).  Here the
// library holds the communicator -- RCCL over xGMI, one process per GPU -- and ENQUEUES the collectives itself on the handle's main stream, between
// the kernels of two phases: a sharded frame is one call again (ifx_owner_process_frame_device), no host round trip inside it, and a C++ host
// (instancefusion_amd/host/ifx_host.hpp) can run sharded without Python.
//
// RCCL is loaded lazily with dlopen (librccl.so.1): the single-GPU product never maps it, libifx.so has no link-time dependency on it, and a
// process that already carries an RCCL (PyTorch ships one) shares it through the SONAME.  The communicator is either created here from a
// ncclUniqueId the host distributes (ifx_comm_unique_id on rank 0 -> every rank ifx_owner_init_comm), or adopted (ifx_owner_set_comm: a ncclComm_t
// the host already owns).
//
// Exchange points of a frame (ifx_owner_exchange; all buffers that travel together are one allocation, so each is ONE collective):
//   after phase 0  key_index                               u64 MIN     8 B / pixel
//   after phase 1  assoc_key (per measurement pixel)       u64 MIN     2 B / pixel    (association by the owner of each candidate: distance bits | window position)
//   after phase 2  key_index                               u64 MIN     8
//   after phase 3  index_tap                               i32 SUM    16             (disjoint supports: the winner's owner writes, the others hold zeros)
//   after phase 4  [key_splat | key_ids]                   u64 MIN    16             (key_both folded in by k_merge_both)
//   after phase 5  [pred_conf | pred_normal | pred_image | pred_inst | pred_time | tail: vote mass]   i32 SUM    30 (+ 16 B)   (the vertex is rebuilt from the key's depth)
// six collectives, 80 B / pixel (round 2: fourteen collectives, 130 B / pixel; start of round 3: six, 122 B).
// Options (round 6): own_lazy_ids -- exchange 4 carries [key_splat | the id keys of the sampled 10 x 10 lattice]: 72 B / pixel, the whole id image's keys travel with a
// segmentation call; own_key_rs -- the index keys of exchanges 0 and 2 as reduce-scatter + all-gather of the creation numbers (op 6): 68 B / pixel in all-reduce-equivalent
// bytes, eight collectives.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "ifx_ctx.h"

namespace {
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    std::string err;
};
Rccl g_rccl;

bool rccl_load(std::string& err)
{
    if (g_rccl.lib) return true;
    if (!g_rccl.err.empty()) { err = g_rccl.err; return false; }
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
    if (!lib) { g_rccl.err = std::string("RCCL not found (dlopen librccl.so.1): ") + (dlerror() ? dlerror() : ""); err = g_rccl.err; return false; }
#define SYM(field, name) g_rccl.field = (decltype(g_rccl.field))dlsym(lib, name); if (!g_rccl.field) { g_rccl.err = std::string("RCCL symbol missing: ") + name; err = g_rccl.err; dlclose(lib); return false; }
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy") SYM(CommCount, "ncclCommCount")
    SYM(CommUserRank, "ncclCommUserRank") SYM(AllReduce, "ncclAllReduce") SYM(AllGather, "ncclAllGather") SYM(ReduceScatter, "ncclReduceScatter") SYM(Broadcast, "ncclBroadcast") SYM(Reduce, "ncclReduce") SYM(GetErrorString, "ncclGetErrorString")
    SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd")
#undef SYM
    g_rccl.lib = lib;
    return true;
}

struct Comm {
    ncclComm_t comm = nullptr;
    bool owned = false;            // created here (destroyed with the handle) or adopted from the host
    long long n_coll = 0, bytes = 0;   // collectives enqueued / bytes reduced since the last reset (ifx_owner_exchange_stats)
    int* d_cnt = nullptr;          // [G] slot counts of the kNN all-gather
    void *knn_send = nullptr, *knn_recv = nullptr, *knn_all = nullptr;
    size_t knn_send_cap = 0, knn_recv_cap = 0, knn_all_cap = 0;
    void *rs_tile = nullptr, *rs_ids = nullptr, *rs_all = nullptr;   // op 6: this rank's tile of reduced keys, its creation numbers, every tile's
    size_t rs_cap = 0;                                                // in keys per tile
};
Comm* comm_of(ifx* h) { return (Comm*)h->comm; }
}   // namespace

#define NCCLCHK(h, call)                                                                                     \
    do {                                                                                                     \
        ncclResult_t r_ = (call);                                                                            \
        if (r_ != ncclSuccess) { (h)->err = std::string(#call) + ": " + g_rccl.GetErrorString(r_); return IFX_E_HIP; } \
    } while (0)

void ifx_comm_free(ifx* h)
{
    Comm* c = comm_of(h);
    if (!c) return;
    if (c->comm && c->owned && g_rccl.lib) g_rccl.CommDestroy(c->comm);
    if (c->d_cnt) hipFree(c->d_cnt);
    if (c->knn_send) hipFree(c->knn_send);
    if (c->knn_recv) hipFree(c->knn_recv);
    if (c->knn_all) hipFree(c->knn_all);
    if (c->rs_tile) hipFree(c->rs_tile);
    if (c->rs_ids) hipFree(c->rs_ids);
    if (c->rs_all) hipFree(c->rs_all);
    delete c;
    h->comm = nullptr;
}

extern "C" int ifx_comm_unique_id(uint8_t* out128)
{
    if (!out128) return IFX_E_INVALID;
    std::string e;
    if (!rccl_load(e)) return IFX_E_STATE;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return IFX_E_HIP;
    memcpy(out128, &id, 128);
    return IFX_OK;
}

static int comm_check(ifx* h, const char* who)
{
    if (!h->own) { h->err = std::string(who) + ": the handle was not created for a sharded map (ifx_config::n_ranks > 1, or -1 for a world of one)"; return IFX_E_STATE; }
    return IFX_OK;
}

extern "C" int ifx_owner_init_comm(ifx_t* h, const uint8_t* unique_id128)
{
    if (!h || !unique_id128) return IFX_E_INVALID;
    int r = comm_check(h, "ifx_owner_init_comm");
    if (r) return r;
    if (!rccl_load(h->err)) return IFX_E_STATE;
    ifx_comm_free(h);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    ncclUniqueId id;
    memcpy(&id, unique_id128, 128);
    Comm* c = new Comm();
    h->comm = c;
    ncclResult_t nr = g_rccl.CommInitRank(&c->comm, h->own_g, id, h->cfg.rank);
    if (nr != ncclSuccess) { h->err = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(nr); ifx_comm_free(h); return IFX_E_HIP; }
    c->owned = true;
    return IFX_OK;
}

extern "C" int ifx_owner_set_comm(ifx_t* h, void* nccl_comm)
{
    if (!h) return IFX_E_INVALID;
    int r = comm_check(h, "ifx_owner_set_comm");
    if (r) return r;
    ifx_comm_free(h);
    if (!nccl_comm) return IFX_OK;   // back to caller-driven exchanges
    if (!rccl_load(h->err)) return IFX_E_STATE;
    int n = 0, me = -1;
    NCCLCHK(h, g_rccl.CommCount((ncclComm_t)nccl_comm, &n));
    NCCLCHK(h, g_rccl.CommUserRank((ncclComm_t)nccl_comm, &me));
    if (n != h->own_g || me != h->cfg.rank) { h->err = "ifx_owner_set_comm: the communicator's size / rank differ from the handle's n_ranks / rank"; return IFX_E_INVALID; }
    Comm* c = new Comm();
    c->comm = (ncclComm_t)nccl_comm;
    h->comm = c;
    return IFX_OK;
}

// op 6 (option own_key_rs): key images whose consumers read the winner's creation number only (the index keys: k_index_resolve names the winner and asks "is it mine?", the depth
// has done its work in the MIN).  The MIN runs as a reduce-scatter over G tiles, then only the low words travel on: 8 + 4 bytes per key and link direction x (G - 1) / G
// instead of the all-reduce's 16.  The buffer comes back with the depth stripped: (u64) creation number, empty keys whole.
__global__ void k_keys_low(const unsigned long long* __restrict__ keys, int n, uint32_t* __restrict__ low)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) low[k] = (uint32_t)(keys[k] & 0xFFFFFFFFull);
}
__global__ void k_keys_from_low(const uint32_t* __restrict__ low, int n, unsigned long long* __restrict__ keys)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) { const uint32_t v = low[k]; keys[k] = v == 0xFFFFFFFFu ? IFX_KEY_EMPTY : (unsigned long long)v; }
}
static int comm_keys_rs(ifx* h, Comm* c, void* ptr, int64_t nbytes)
{
    const int G = h->own_g, n = (int)(nbytes / 8), tile = (n + G - 1) / G;
    if ((size_t)tile > c->rs_cap) {
        if (c->rs_tile) hipFree(c->rs_tile);
        if (c->rs_ids) hipFree(c->rs_ids);
        if (c->rs_all) hipFree(c->rs_all);
        c->rs_tile = c->rs_ids = c->rs_all = nullptr; c->rs_cap = 0;
        HIPCHK(h, hipMalloc(&c->rs_tile, (size_t)tile * 8)); HIPCHK(h, hipMalloc(&c->rs_ids, (size_t)tile * 4)); HIPCHK(h, hipMalloc(&c->rs_all, (size_t)tile * 4 * G));
        c->rs_cap = (size_t)tile;
    }
    // (G * tile - n < G keys are read past the image: the allocation carries that slack, IFX_KEY_SLACK, and nothing of it comes back)
    NCCLCHK(h, g_rccl.ReduceScatter(ptr, c->rs_tile, (size_t)tile, ncclUint64, ncclMin, c->comm, h->stream));
    hipLaunchKernelGGL(k_keys_low, dim3((tile + 255) / 256), dim3(256), 0, h->stream, (const unsigned long long*)c->rs_tile, tile, (uint32_t*)c->rs_ids);
    NCCLCHK(h, g_rccl.AllGather(c->rs_ids, c->rs_all, (size_t)tile, ncclUint32, c->comm, h->stream));
    hipLaunchKernelGGL(k_keys_from_low, dim3((n + 255) / 256), dim3(256), 0, h->stream, (const uint32_t*)c->rs_all, n, (unsigned long long*)ptr);
    c->n_coll += 1;                       // (two collectives at this exchange point: ifx_comm_exchange counts the other)
    c->bytes += (nbytes + nbytes / 2) / 2;   // in all-reduce-equivalent bytes, the unit of the other exchanges (a ring all-reduce of S bytes moves 2 S (G - 1) / G per link): (8 + 4) / 2 per key
    return IFX_OK;
}
// one buffer of an exchange point
static int comm_one(ifx* h, Comm* c, void* ptr, int64_t nbytes, int32_t opc)
{
    // ops of ifx_owner_exchange: 0 unsigned 64-bit MIN (keys: depth | creation number), 1 int32 SUM (disjoint supports: a bitwise merge), 2 int32 MIN, 3 int32 MAX,
    // 4 | root << 8 broadcast, 5 | root << 8 int32 SUM to the root only, 6 unsigned 64-bit MIN of which only the low words are wanted back (comm_keys_rs)
    if ((opc & 0xFF) == 6) return comm_keys_rs(h, c, ptr, nbytes);
    if ((opc & 0xFF) == 4) {   // broadcast from rank ops >> 8 (the pose block of a frame tracked by one rank)
        NCCLCHK(h, g_rccl.Broadcast(ptr, ptr, (size_t)nbytes, ncclInt8, opc >> 8, c->comm, h->stream));
        c->bytes += nbytes;
        return IFX_OK;
    }
    if ((opc & 0xFF) == 5) {   // int32 SUM reduced to rank ops >> 8 only (the prediction of a camera that one rank tracks)
        NCCLCHK(h, g_rccl.Reduce(ptr, ptr, (size_t)nbytes / 4, ncclInt32, ncclSum, opc >> 8, c->comm, h->stream));
        c->bytes += nbytes;
        return IFX_OK;
    }
    const bool u64 = opc == 0;
    const ncclRedOp_t op = opc == 0 ? ncclMin : (opc == 1 ? ncclSum : (opc == 2 ? ncclMin : ncclMax));
    NCCLCHK(h, g_rccl.AllReduce(ptr, ptr, (size_t)nbytes / (u64 ? 8 : 4), u64 ? ncclUint64 : ncclInt32, op, c->comm, h->stream));
    c->bytes += nbytes;
    return IFX_OK;
}
// the exchange after phase `phase` (or 200: the pending exchange point of a segmentation call), enqueued on the handle's main stream
int ifx_comm_exchange(ifx* h, int phase)
{
    Comm* c = comm_of(h);
    if (!c || !c->comm) { h->err = "no communicator: ifx_owner_init_comm / ifx_owner_set_comm first"; return IFX_E_STATE; }
    void* ptrs[8]; int64_t bytes[8]; int32_t ops[8];
    const int n = ifx_owner_exchange(h, phase, ptrs, bytes, ops, 8);
    if (n < 0) return n;
    // the buffers of one exchange point travel as ONE group (the keys and the 8-byte "surfel 0" word of exchanges 0 / 4; the prediction block and its tail of
    // exchange 5; the two statistics of a segmentation call): aggregated by RCCL into one launch
    const bool group = n > 1;   // (op 6 has kernels between its two collectives: such a buffer is the only one of its exchange point)
    if (group) NCCLCHK(h, g_rccl.GroupStart());
    int rc = IFX_OK;
    for (int k = 0; k < n && k < 8 && rc == IFX_OK; k++) rc = comm_one(h, c, ptrs[k], bytes[k], ops[k]);
    if (group) NCCLCHK(h, g_rccl.GroupEnd());
    if (n > 0) c->n_coll++;   // (one exchange point = one group = one RCCL launch)
    return rc;
}
int ifx_comm_ready(ifx* h) { Comm* c = comm_of(h); return c && c->comm; }
// option own_track_rows: the exact accumulator rows of a tracker iteration, summed over the ranks (f64 SUM of grid-valued terms: exact, hence order-independent) -- on the
// stream the tracker run is being enqueued on
int ifx_comm_allreduce_f64(ifx* h, double* d_ptr, int n)
{
    Comm* c = comm_of(h);
    if (!c || !c->comm) { h->err = "no communicator: ifx_owner_init_comm / ifx_owner_set_comm first"; return IFX_E_STATE; }
    NCCLCHK(h, g_rccl.AllReduce(d_ptr, d_ptr, (size_t)n, ncclDouble, ncclSum, c->comm, h->cur));
    c->n_coll++; c->bytes += (long long)n * 8;
    return IFX_OK;
}

// ranks of the communicator the handle's collectives run on, as RCCL counts them (ncclCommCount); 0: no communicator yet
extern "C" int ifx_owner_comm_ranks(ifx_t* h)
{
    if (!h) return IFX_E_INVALID;
    Comm* c = comm_of(h);
    if (!c || !c->comm) return 0;
    int n = 0;
    NCCLCHK(h, g_rccl.CommCount(c->comm, &n));
    return n;
}

extern "C" int ifx_owner_exchange_stats(ifx_t* h, int64_t* out2, int reset)
{
    if (!h || !out2) return IFX_E_INVALID;
    Comm* c = comm_of(h);
    out2[0] = c ? c->n_coll : 0; out2[1] = c ? c->bytes : 0;
    if (c && reset) { c->n_coll = 0; c->bytes = 0; }
    return IFX_OK;
}

// InstanceFusion::flannKnnVoteSurfelMap on the sharded map, the all-gather inside the library: every rank's slots (x, y, z, creation number; label)
// gathered in rank order -- 20 B per slot, once per smoothing (every > 40 frames) --, then the grid search for the surfels of this rank.
extern "C" int ifx_owner_knn_vote_colour(ifx_t* h)
{
    if (!h) return IFX_E_INVALID;
    int r = comm_check(h, "ifx_owner_knn_vote_colour");
    if (r) return r;
    Comm* c = comm_of(h);
    if (!c || !c->comm) { h->err = "no communicator: ifx_owner_init_comm / ifx_owner_set_comm first"; return IFX_E_STATE; }
    const int G = h->own_g, me = h->cfg.rank;
    void *d_pts = nullptr, *d_lab = nullptr;
    int n = 0;
    r = ifx_owner_knn_export(h, &d_pts, &d_lab, &n);
    if (r) return r;
    if (!c->d_cnt) HIPCHK(h, hipMalloc(&c->d_cnt, (size_t)(G + 1) * 4));
    HIPCHK(h, hipMemcpyAsync(c->d_cnt + G, &n, 4, hipMemcpyHostToDevice, h->stream));
    NCCLCHK(h, g_rccl.AllGather(c->d_cnt + G, c->d_cnt, 1, ncclInt32, c->comm, h->stream));
    std::vector<int> cnt(G);
    HIPCHK(h, hipMemcpyAsync(cnt.data(), c->d_cnt, (size_t)G * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    int nmax = 0; long long total = 0;
    for (int g = 0; g < G; g++) { nmax = std::max(nmax, cnt[g]); total += cnt[g]; }
    if (total == 0) return IFX_OK;
    // equal-sized contributions (ncclAllGather): [nmax float4 | nmax int32] per rank, unused rows are never read
    const size_t row = (size_t)nmax * 20;
    auto grow = [&](void*& p, size_t& cap, size_t need) -> hipError_t { if (need <= cap) return hipSuccess; if (p) hipFree(p); p = nullptr; cap = 0; hipError_t e = hipMalloc(&p, need); if (e == hipSuccess) cap = need; return e; };
    HIPCHK(h, grow(c->knn_send, c->knn_send_cap, row));
    HIPCHK(h, grow(c->knn_recv, c->knn_recv_cap, row * G));
    HIPCHK(h, grow(c->knn_all, c->knn_all_cap, (size_t)total * 20));
    if (n > 0) {
        HIPCHK(h, hipMemcpyAsync(c->knn_send, d_pts, (size_t)n * 16, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync((uint8_t*)c->knn_send + (size_t)nmax * 16, d_lab, (size_t)n * 4, hipMemcpyDeviceToDevice, h->stream));
    }
    NCCLCHK(h, g_rccl.AllGather(c->knn_send, c->knn_recv, row, ncclInt8, c->comm, h->stream));
    c->n_coll += 2; c->bytes += (long long)row * G;
    uint8_t* allp = (uint8_t*)c->knn_all;
    uint8_t* alll = allp + (size_t)total * 16;
    long long off = 0, my_off = 0;
    for (int g = 0; g < G; g++) {
        if (g == me) my_off = off;
        if (cnt[g] > 0) {
            HIPCHK(h, hipMemcpyAsync(allp + (size_t)off * 16, (uint8_t*)c->knn_recv + row * g, (size_t)cnt[g] * 16, hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(h, hipMemcpyAsync(alll + (size_t)off * 4, (uint8_t*)c->knn_recv + row * g + (size_t)nmax * 16, (size_t)cnt[g] * 4, hipMemcpyDeviceToDevice, h->stream));
        }
        off += cnt[g];
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ifx_owner_knn_vote(h, allp, alll, (int)total, (int)my_off);
}

// InstanceFusion::processInstance on the sharded map in one call: the begin / resume state machine of ifx_instance.hip with the exchanges done here.
// flags bit 0 (kNN smoothing of the colours) runs ifx_owner_knn_vote_colour after the call.
extern "C" int ifx_owner_process_segmentation(ifx_t* h, const uint8_t* rgb, const uint16_t* depth, const uint8_t* masks, const int32_t* class_ids, int nm, int frame, int flags)
{
    if (!h) return IFX_E_INVALID;
    int r = comm_check(h, "ifx_owner_process_segmentation");
    if (r) return r;
    if (!ifx_comm_ready(h)) { h->err = "no communicator: ifx_owner_init_comm / ifx_owner_set_comm first"; return IFX_E_STATE; }
    r = ifx_owner_segmentation_begin(h, rgb, depth, masks, class_ids, nm, frame, flags & ~1);
    while (r == 1) {
        r = ifx_comm_exchange(h, 200);
        if (r) { h->oseg_state = 0; h->oseg_pending = 0; return r; }   // a failed call is over: the next one starts from scratch
        r = ifx_owner_segmentation_resume(h);
    }
    if (r) return r;
    if (flags & 1) return ifx_owner_knn_vote_colour(h);
    return IFX_OK;
}
