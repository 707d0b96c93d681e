// statmc_rccl.hip -- the halo exchange of a sharded film between PROCESSES (one process per GPU), over RCCL, behind the C ABI.
//
// New capability (the reference is single-GPU: SURVEY.md 8e); `north_star`: "film tiles shard across the 8 GPUs of one node with an
// RCCL halo exchange over xGMI for the filter-window overlap ... via a thin C-ABI HIP shim".  statmc_halo_exchange (statmc_abi.hip)
// is the same exchange for ONE process that owns every device (device-to-device copies); the Python ranks reach RCCL through
// torch.distributed (statmc_amd/sharding.py).  This file gives a multi-process C or C++ host the entry point: every rank packs its
// block (statmc_prepass_pack) and calls statmc_halo_exchange_rccl on its own block + halo image.
//
// Shape of the exchange: point to point, every neighbour one hop, no ring, no collective -- xGMI is point-to-point, and a halo is
// one message per side.  Two phases, as in statmc_halo_exchange: the columns of the owned rows first (left / right neighbours),
// then the rows over the widened block so that the corners ride along (upper / lower neighbours); each phase is one RCCL group of
// at most two sends and two receives on the block's stream, so the second phase's sends are ordered behind the first phase's
// receives by the stream.  Row halos are contiguous in a packed-row image and go straight from / into the block + halo image; column
// halos pass through a staging buffer (hipMemcpy2DAsync on the same stream).
//
// RCCL is bound at RUN time (dlsym on what the process already has loaded -- the library that made the communicator --, else
// dlopen of librccl.so.1): libstatmc_hip.so has no link-time dependency on it, and a host that never shards never loads it.
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>

#include <map>
#include <mutex>

#include <rccl/rccl.h>

#include "statmc_device.h"

namespace {

struct Rccl {
    bool tried = false, ok = false;
    const char *why = "";
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
std::mutex g_rccl_mu;
Rccl g_rccl;

template <class F>
bool bind(void *handle, const char *name, F &fn) {
    void *p = dlsym(handle ? handle : RTLD_DEFAULT, name);
    fn = reinterpret_cast<F>(p);
    return p != nullptr;
}

const Rccl &rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    Rccl &R = g_rccl;
    if (R.tried) return R;
    R.tried = true;
    // the library the process already has (a host that linked RCCL, or a Python process whose torch brought its own copy): the
    // communicator handed to us was made by THAT library
    void *handle = nullptr;
    if (dlsym(RTLD_DEFAULT, "ncclSend") == nullptr) {
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
        }
        if (!handle) {
            R.why = "librccl.so.1 not found (dlopen)";
            return R;
        }
    }
    R.ok = bind(handle, "ncclGetUniqueId", R.GetUniqueId) && bind(handle, "ncclCommInitRank", R.CommInitRank) &&
           bind(handle, "ncclCommDestroy", R.CommDestroy) && bind(handle, "ncclCommCount", R.CommCount) &&
           bind(handle, "ncclCommUserRank", R.CommUserRank) && bind(handle, "ncclGroupStart", R.GroupStart) &&
           bind(handle, "ncclGroupEnd", R.GroupEnd) && bind(handle, "ncclSend", R.Send) && bind(handle, "ncclRecv", R.Recv) &&
           bind(handle, "ncclGetErrorString", R.GetErrorString);
    if (!R.ok) R.why = "the RCCL library lacks one of the entry points used";
    return R;
}

// staging for the column halos: four buffers (send left / right, receive left / right) per (device, stream), grown on demand
struct Staging {
    char *ptr = nullptr;
    size_t bytes = 0;
};
std::mutex g_stage_mu;
std::map<std::pair<int, void *>, Staging> g_stage;

int staging(int dev, void *stream, size_t bytes, char **out) {
    std::lock_guard<std::mutex> lk(g_stage_mu);
    Staging &s = g_stage[{dev, stream}];
    if (s.bytes < bytes) {
        if (s.ptr) {
            if (hipError_t e = hipStreamSynchronize(static_cast<hipStream_t>(stream)); e != hipSuccess)
                return statmc::abi_fail(STATMC_ERR_HIP, "hipStreamSynchronize: %s", hipGetErrorString(e));
            (void)hipFree(s.ptr);
        }
        s.ptr = nullptr;
        s.bytes = 0;
        if (hipError_t e = hipMalloc(reinterpret_cast<void **>(&s.ptr), bytes); e != hipSuccess)
            return statmc::abi_fail(STATMC_ERR_HIP, "hipMalloc (halo staging): %s", hipGetErrorString(e));
        s.bytes = bytes;
    }
    *out = s.ptr;
    return STATMC_OK;
}

int channels_of(const statmc_image &im) {   // fp32 channels per pixel of a packed-row image, 0 if the pitch is not a whole pixel count
    if (im.cols <= 0 || im.step % ((size_t)im.cols * 4) != 0) return 0;
    return (int)(im.step / ((size_t)im.cols * 4));
}

}  // namespace

extern "C" {

int statmc_rccl_available(void) { return rccl().ok ? 1 : 0; }

int statmc_rccl_unique_id(void *id128) {
    if (!id128) return statmc::abi_fail(STATMC_ERR_INVALID, "null id");
    const Rccl &R = rccl();
    if (!R.ok) return statmc::abi_fail(STATMC_ERR_UNSUPPORTED, "RCCL: %s", R.why);
    static_assert(sizeof(ncclUniqueId) == 128, "statmc.h documents a 128-byte id");
    ncclUniqueId id;
    if (ncclResult_t e = R.GetUniqueId(&id); e != ncclSuccess) return statmc::abi_fail(STATMC_ERR_HIP, "ncclGetUniqueId: %s", R.GetErrorString(e));
    memcpy(id128, &id, sizeof(id));
    return STATMC_OK;
}

int statmc_rccl_comm_create(void **comm, int n_ranks, int rank, const void *id128) {
    if (!comm || !id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return statmc::abi_fail(STATMC_ERR_INVALID, "bad communicator arguments");
    const Rccl &R = rccl();
    if (!R.ok) return statmc::abi_fail(STATMC_ERR_UNSUPPORTED, "RCCL: %s", R.why);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    if (ncclResult_t e = R.CommInitRank(&c, n_ranks, id, rank); e != ncclSuccess)
        return statmc::abi_fail(STATMC_ERR_HIP, "ncclCommInitRank (rank %d of %d): %s", rank, n_ranks, R.GetErrorString(e));
    *comm = c;
    return STATMC_OK;
}

int statmc_rccl_comm_destroy(void *comm) {
    {   // the column-halo staging buffers of the current device go with the communicator (they are per (device, stream), grown on demand)
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) {
            std::lock_guard<std::mutex> lk(g_stage_mu);
            for (auto it = g_stage.begin(); it != g_stage.end();) {
                if (it->first.first == dev) {
                    if (it->second.ptr) {
                        (void)hipStreamSynchronize(static_cast<hipStream_t>(it->first.second));
                        (void)hipFree(it->second.ptr);
                    }
                    it = g_stage.erase(it);
                } else {
                    ++it;
                }
            }
        }
        (void)hipGetLastError();
    }
    if (!comm) return STATMC_OK;
    const Rccl &R = rccl();
    if (!R.ok) return statmc::abi_fail(STATMC_ERR_UNSUPPORTED, "RCCL: %s", R.why);
    if (ncclResult_t e = R.CommDestroy(static_cast<ncclComm_t>(comm)); e != ncclSuccess)
        return statmc::abi_fail(STATMC_ERR_HIP, "ncclCommDestroy: %s", R.GetErrorString(e));
    return STATMC_OK;
}

int statmc_halo_exchange_rccl(const statmc_block *block, int gx, int gy, int block_w, int block_h, int radius, void *nccl_comm, int rank) {
    if (!block || gx < 1 || gy < 1 || block_w < 1 || block_h < 1 || radius < 0) return statmc::abi_fail(STATMC_ERR_INVALID, "bad block grid");
    if ((gx > 1 && block_w < radius) || (gy > 1 && block_h < radius))
        return statmc::abi_fail(STATMC_ERR_INVALID, "block %dx%d smaller than the radius %d: a halo comes from one ring of neighbours", block_w, block_h, radius);
    const int n = gx * gy, r = radius;
    if (rank < 0 || rank >= n) return statmc::abi_fail(STATMC_ERR_INVALID, "rank %d outside the %dx%d grid", rank, gx, gy);
    if (n == 1 || r == 0) return STATMC_OK;
    if (!nccl_comm) return statmc::abi_fail(STATMC_ERR_INVALID, "null communicator");
    const Rccl &R = rccl();
    if (!R.ok) return statmc::abi_fail(STATMC_ERR_UNSUPPORTED, "RCCL: %s", R.why);
    ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
    int count = 0, me = -1;
    if (R.CommCount(comm, &count) != ncclSuccess || R.CommUserRank(comm, &me) != ncclSuccess || count != n || me != rank)
        return statmc::abi_fail(STATMC_ERR_INVALID, "the communicator has %d ranks and calls this one %d: expected %d ranks, rank %d (rank = by * gx + bx)", count, me, n, rank);
    const int bx = rank % gx, by = rank / gx;
    const int pl = bx > 0 ? r : 0, pr = bx + 1 < gx ? r : 0, pt = by > 0 ? r : 0, pb = by + 1 < gy ? r : 0;
    const statmc_image &im = block->packed;
    const int pch = channels_of(im);
    if (!im.data || pch == 0 || im.cols != block_w + pl + pr || im.rows != block_h + pt + pb)
        return statmc::abi_fail(STATMC_ERR_INVALID, "the packed image is not the %dx%d block + halo image of this rank (packed rows, fp32 channels)",
                                block_w + pl + pr, block_h + pt + pb);
    const size_t px = (size_t)pch * 4;
    hipStream_t s = static_cast<hipStream_t>(block->stream);
    int cur = 0;
    if (hipError_t e = hipGetDevice(&cur); e != hipSuccess) return statmc::abi_fail(STATMC_ERR_HIP, "hipGetDevice: %s", hipGetErrorString(e));
    if (cur != block->device) {
        if (hipError_t e = hipSetDevice(block->device); e != hipSuccess) return statmc::abi_fail(STATMC_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    }
    int rc = STATMC_OK;
    auto hip = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == STATMC_OK) rc = statmc::abi_fail(STATMC_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    };
    auto nc = [&](ncclResult_t e, const char *what) {
        if (e != ncclSuccess && rc == STATMC_OK) rc = statmc::abi_fail(STATMC_ERR_HIP, "%s: %s", what, R.GetErrorString(e));
    };
    char *base = static_cast<char *>(im.data);
    auto at = [&](int x, int y) { return base + (size_t)y * im.step + (size_t)x * px; };

    // ---- phase 1: columns of the owned rows (r x block_h pixels per side), through the staging buffer
    if (pl || pr) {
        const size_t side = (size_t)r * block_h * px;
        char *stage = nullptr;
        rc = staging(block->device, block->stream, 4 * side, &stage);
        char *send_l = stage, *send_r = stage + side, *recv_l = stage + 2 * side, *recv_r = stage + 3 * side;
        if (rc == STATMC_OK && pl) hip(hipMemcpy2DAsync(send_l, (size_t)r * px, at(pl, pt), im.step, (size_t)r * px, block_h, hipMemcpyDeviceToDevice, s), "hipMemcpy2DAsync");
        if (rc == STATMC_OK && pr) hip(hipMemcpy2DAsync(send_r, (size_t)r * px, at(pl + block_w - r, pt), im.step, (size_t)r * px, block_h, hipMemcpyDeviceToDevice, s), "hipMemcpy2DAsync");
        if (rc == STATMC_OK) {
            nc(R.GroupStart(), "ncclGroupStart");
            if (pl) { nc(R.Send(send_l, side, ncclChar, rank - 1, comm, s), "ncclSend"); nc(R.Recv(recv_l, side, ncclChar, rank - 1, comm, s), "ncclRecv"); }
            if (pr) { nc(R.Send(send_r, side, ncclChar, rank + 1, comm, s), "ncclSend"); nc(R.Recv(recv_r, side, ncclChar, rank + 1, comm, s), "ncclRecv"); }
            nc(R.GroupEnd(), "ncclGroupEnd");
        }
        if (rc == STATMC_OK && pl) hip(hipMemcpy2DAsync(at(0, pt), im.step, recv_l, (size_t)r * px, (size_t)r * px, block_h, hipMemcpyDeviceToDevice, s), "hipMemcpy2DAsync");
        if (rc == STATMC_OK && pr) hip(hipMemcpy2DAsync(at(pl + block_w, pt), im.step, recv_r, (size_t)r * px, (size_t)r * px, block_h, hipMemcpyDeviceToDevice, s), "hipMemcpy2DAsync");
    }
    // ---- phase 2: rows over the widened width (r rows of packed pixels: contiguous, straight out of / into the image)
    if (rc == STATMC_OK && (pt || pb)) {
        const size_t rows = (size_t)r * im.step;
        nc(R.GroupStart(), "ncclGroupStart");
        if (pt) { nc(R.Send(at(0, pt), rows, ncclChar, rank - gx, comm, s), "ncclSend"); nc(R.Recv(at(0, 0), rows, ncclChar, rank - gx, comm, s), "ncclRecv"); }
        if (pb) { nc(R.Send(at(0, pt + block_h - r), rows, ncclChar, rank + gx, comm, s), "ncclSend"); nc(R.Recv(at(0, pt + block_h), rows, ncclChar, rank + gx, comm, s), "ncclRecv"); }
        nc(R.GroupEnd(), "ncclGroupEnd");
    }
    if (cur != block->device) (void)hipSetDevice(cur);
    return rc;
}

}  // extern "C"
