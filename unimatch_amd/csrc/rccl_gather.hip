// Collecting the per-rank predictions: RCCL all-gather behind the C ABI.
//
// The hot path shards by sample with no data-path collective; the one exchange is a single ncclAllGather of the final
// prediction ([B/N, V, H, W] fp32, 3.1 MB per pair at 512x768) on a stream the caller chooses.  The reference has no
// inference-time collective (its process-group bring-up, utils/dist_utils.py:12-30, serves DDP training); this file takes
// that role without torch: the communicator is bootstrapped either from a 128-byte unique id the caller distributes
// (um_comm_unique_id + um_comm_init_rank) or through a file on a filesystem all ranks of the node see (um_comm_init_file).
//
// RCCL is bound lazily by soname (dlopen "librccl.so.1"): the library has no link-time RCCL dependency, and inside a
// process that already loaded PyTorch's RCCL the same copy is reused instead of a second one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <mutex>
#include "../../include/unimatch_hip.h"

extern void um_set_error(const char* fmt, ...);

namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::once_flag g_once;
char g_load_error[256] = "";

void load_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) {
        snprintf(g_load_error, sizeof(g_load_error), "cannot dlopen librccl.so.1: %s", dlerror());
        return;
    }
    bool ok = true;
    auto sym = [&](const char* name) {
        void* p = dlsym(g_rccl.handle, name);
        if (!p) {
            snprintf(g_load_error, sizeof(g_load_error), "librccl lacks %s", name);
            ok = false;
        }
        return p;
    };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))sym("ncclCommCount");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    if (!ok) {
        dlclose(g_rccl.handle);
        g_rccl.handle = nullptr;
    }
}

bool have_rccl() {
    std::call_once(g_once, load_rccl);
    if (!g_rccl.handle) um_set_error("RCCL unavailable: %s", g_load_error);
    return g_rccl.handle != nullptr;
}

int fail(const char* what, ncclResult_t r) {
    um_set_error("%s: RCCL error %d (%s)", what, (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return UM_ERR_COLLECTIVE;
}
}  // namespace

extern "C" int um_comm_unique_id(void* id_out) {
    if (!id_out) {
        um_set_error("um_comm_unique_id: null output");
        return UM_ERR_BAD_ARG;
    }
    if (!have_rccl()) return UM_ERR_UNSUPPORTED;
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return fail("ncclGetUniqueId", r);
    memcpy(id_out, id.internal, UM_COMM_ID_BYTES);
    return 0;
}

extern "C" int um_comm_init_rank(void** comm_out, const void* id, int rank, int world) {
    if (!comm_out || !id || world <= 0 || rank < 0 || rank >= world) {
        um_set_error("um_comm_init_rank: null pointer or rank %d outside [0, %d)", rank, world);
        return UM_ERR_BAD_ARG;
    }
    if (!have_rccl()) return UM_ERR_UNSUPPORTED;
    ncclUniqueId uid;
    memcpy(uid.internal, id, UM_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    ncclResult_t r = g_rccl.CommInitRank(&comm, world, uid, rank);     // binds the calling thread's current HIP device
    if (r != ncclSuccess) return fail("ncclCommInitRank", r);
    *comm_out = (void*)comm;
    return 0;
}

// File bootstrap: rank 0 removes whatever is at `path`, writes {magic, world, wall-clock stamp, id} to `<path>.tmp` and renames it
// to `path` (atomic on one filesystem); the other ranks poll for a record with the right magic and world that is younger than
// UM_ID_FILE_MAX_AGE seconds (a file left behind by a crashed job of another day is ignored instead of handing out a dead id),
// and rank 0 deletes the file once ncclCommInitRank has returned, i.e. after every rank has joined.  Two jobs must not share a
// path at the same time: key it by job (unimatch_amd.dist.job_id_file: rendezvous port + launcher pid).  A job that crashed
// between publish and unlink leaves a record that is still "fresh" for UM_ID_FILE_MAX_AGE seconds: um_comm_init_file_nonce
// stores a caller-chosen per-job nonce in the record and its readers skip records of another nonce, so a relaunch under the
// same path never joins the dead id (um_comm_init_file = nonce 0 on both sides).
#define UM_ID_FILE_MAX_AGE 600
namespace {
struct IdRecord {
    char magic[8];
    int world;
    int nonce;                  // per-job tag chosen by the caller (0: um_comm_init_file)
    long long stamp;
    unsigned char id[UM_COMM_ID_BYTES];
};
const char kIdMagic[8] = {'U', 'M', 'R', 'C', 'C', 'L', '0', '2'};
}  // namespace

extern "C" int um_comm_init_file_nonce(void** comm_out, const char* path, int rank, int world, int timeout_seconds, int nonce);

extern "C" int um_comm_init_file(void** comm_out, const char* path, int rank, int world, int timeout_seconds) {
    return um_comm_init_file_nonce(comm_out, path, rank, world, timeout_seconds, 0);
}

extern "C" int um_comm_init_file_nonce(void** comm_out, const char* path, int rank, int world, int timeout_seconds, int nonce) {
    if (!comm_out || !path || !*path || world <= 0 || rank < 0 || rank >= world) {
        um_set_error("um_comm_init_file: null pointer / empty path or rank %d outside [0, %d)", rank, world);
        return UM_ERR_BAD_ARG;
    }
    if (strlen(path) > 1000) {
        um_set_error("um_comm_init_file: path longer than 1000 bytes");
        return UM_ERR_BAD_ARG;
    }
    IdRecord rec;
    if (rank == 0) {
        memset(&rec, 0, sizeof(rec));
        memcpy(rec.magic, kIdMagic, sizeof(kIdMagic));
        rec.world = world;
        rec.nonce = nonce;
        rec.stamp = (long long)time(nullptr);
        if (int e = um_comm_unique_id(rec.id)) return e;
        (void)unlink(path);                                         // a leftover of an earlier job under the same key
        char tmp[1024 + 8];
        snprintf(tmp, sizeof(tmp), "%s.tmp", path);
        FILE* f = fopen(tmp, "wb");
        if (!f || fwrite(&rec, 1, sizeof(rec), f) != sizeof(rec) || fclose(f) != 0 || rename(tmp, path) != 0) {
            um_set_error("um_comm_init_file: cannot publish the unique id at %s", path);
            return UM_ERR_COLLECTIVE;
        }
    } else {
        const time_t t0 = time(nullptr);
        for (;;) {
            FILE* f = fopen(path, "rb");
            if (f) {
                const size_t n = fread(&rec, 1, sizeof(rec), f);
                fclose(f);
                if (n == sizeof(rec) && memcmp(rec.magic, kIdMagic, sizeof(kIdMagic)) == 0 && rec.world == world &&
                    rec.nonce == nonce && (long long)time(nullptr) - rec.stamp <= UM_ID_FILE_MAX_AGE)
                    break;
            }
            if (timeout_seconds >= 0 && time(nullptr) - t0 > timeout_seconds) {
                um_set_error("um_comm_init_file: rank %d waited %d s for a fresh id record of this job (nonce %d) at %s", rank, timeout_seconds,
                             nonce, path);
                return UM_ERR_COLLECTIVE;
            }
            usleep(20000);
        }
    }
    const int e = um_comm_init_rank(comm_out, rec.id, rank, world);
    if (rank == 0) (void)unlink(path);                              // every rank has joined (or the bootstrap failed): single use
    return e;
}

extern "C" int um_comm_world(void* comm) {
    if (!comm || !have_rccl()) return UM_ERR_BAD_ARG;
    int n = 0;
    ncclResult_t r = g_rccl.CommCount((ncclComm_t)comm, &n);
    return r == ncclSuccess ? n : fail("ncclCommCount", r);
}

extern "C" int um_comm_destroy(void* comm) {
    if (!comm) return 0;
    if (!have_rccl()) return UM_ERR_UNSUPPORTED;
    ncclResult_t r = g_rccl.CommDestroy((ncclComm_t)comm);
    return r == ncclSuccess ? 0 : fail("ncclCommDestroy", r);
}

extern "C" int um_allgather_preds(void* comm, const float* send, float* recv, size_t count_per_rank, void* stream) {
    if (!comm || !send || !recv || count_per_rank == 0) {
        um_set_error("um_allgather_preds: null communicator / buffer or zero count");
        return UM_ERR_BAD_ARG;
    }
    if (!have_rccl()) return UM_ERR_UNSUPPORTED;
    ncclResult_t r = g_rccl.AllGather(send, recv, count_per_rank, ncclFloat32, (ncclComm_t)comm, (hipStream_t)stream);
    return r == ncclSuccess ? 0 : fail("ncclAllGather", r);
}
