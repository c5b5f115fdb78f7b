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

// File bootstrap, three steps so that NO rank enters ncclCommInitRank (which cannot time out) unless all of them will:
//   1. publish   rank 0 removes whatever is at `path`, writes {magic, world, nonce, wall-clock stamp, id} to `<path>.tmp` and renames it
//                to `path` (atomic on one filesystem);
//   2. ack       every other rank polls for a record with the right magic, world and NONCE that is younger than UM_ID_FILE_MAX_AGE
//                seconds, then creates `<path>.ack<rank>`; rank 0 polls for the world - 1 acks;
//   3. go        rank 0 creates `<path>.go`; the others poll for it; everybody calls ncclCommInitRank; rank 0 removes the files.
// Every wait honours `timeout_seconds` on EVERY rank, rank 0 included (round 4's protocol left rank 0 inside ncclCommInitRank for ever
// when a reader never found its record): a rank that is missing, or that disagrees about the nonce, makes all ranks return
// UM_ERR_COLLECTIVE after the timeout instead of hanging one of them.
// Two jobs must not share a path at the same time: key it by job.  The nonce guards the one case a path cannot: a job that crashed
// between publish and clean-up leaves a record that is still "fresh" for UM_ID_FILE_MAX_AGE seconds; a relaunch under the same path
// with another nonce never joins the dead id.  The nonce must be derived from values EVERY rank of the job shares (the path itself,
// MASTER_ADDR:MASTER_PORT, an explicit UM_RCCL_NONCE) -- never from a pid (unimatch_amd.dist.job_nonce).
#define UM_ID_FILE_MAX_AGE 600
namespace {
struct IdRecord {
    char magic[8];
    int world;
    int nonce;                  // per-job tag chosen by the caller (0: um_comm_init_file)
    long long stamp;
    unsigned char id[UM_COMM_ID_BYTES];
};
const char kIdMagic[8] = {'U', 'M', 'R', 'C', 'C', 'L', '0', '3'};      // 03: the ack / go rendezvous (02: publish + poll only)

bool file_exists(const char* p) { return access(p, F_OK) == 0; }
bool touch(const char* p) {
    FILE* f = fopen(p, "wb");
    if (!f) return false;
    fclose(f);
    return true;
}
bool expired(time_t t0, int timeout_seconds) { return timeout_seconds >= 0 && time(nullptr) - t0 > timeout_seconds; }
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
    char aux[1024 + 32];
    auto ack_path = [&](int r) { snprintf(aux, sizeof(aux), "%s.ack%d", path, r); return aux; };
    auto go_path = [&]() { snprintf(aux, sizeof(aux), "%s.go", path); return aux; };
    auto cleanup = [&]() {                                          // rank 0 only
        (void)unlink(path);
        (void)unlink(go_path());
        for (int r = 1; r < world; ++r) (void)unlink(ack_path(r));
    };
    const time_t t0 = time(nullptr);
    IdRecord rec;
    if (rank == 0) {
        memset(&rec, 0, sizeof(rec));
        memcpy(rec.magic, kIdMagic, sizeof(kIdMagic));
        rec.world = world;
        rec.nonce = nonce;
        rec.stamp = (long long)time(nullptr);
        if (int e = um_comm_unique_id(rec.id)) return e;
        cleanup();                                                  // leftovers of an earlier job under the same key
        char tmp[1024 + 8];
        snprintf(tmp, sizeof(tmp), "%s.tmp", path);
        FILE* f = fopen(tmp, "wb");
        if (!f || fwrite(&rec, 1, sizeof(rec), f) != sizeof(rec) || fclose(f) != 0 || rename(tmp, path) != 0) {
            um_set_error("um_comm_init_file: cannot publish the unique id at %s", path);
            return UM_ERR_COLLECTIVE;
        }
        for (int r = 1; r < world; ++r) {
            while (!file_exists(ack_path(r))) {
                if (expired(t0, timeout_seconds)) {
                    um_set_error("um_comm_init_file: rank 0 waited %d s for rank %d to acknowledge the id record (nonce %d) at %s -- is that rank "
                                 "running, on the same filesystem, with the same nonce?", timeout_seconds, r, nonce, path);
                    cleanup();
                    return UM_ERR_COLLECTIVE;
                }
                usleep(20000);
            }
        }
        if (!touch(go_path())) {
            um_set_error("um_comm_init_file: cannot create %s.go", path);
            cleanup();
            return UM_ERR_COLLECTIVE;
        }
    } else {
        for (;;) {
            FILE* f = fopen(path, "rb");
            if (f) {
                const size_t n = fread(&rec, 1, sizeof(rec), f);
                fclose(f);
                if (n == sizeof(rec) && memcmp(rec.magic, kIdMagic, sizeof(kIdMagic)) == 0 && rec.world == world &&
                    rec.nonce == nonce && (long long)time(nullptr) - rec.stamp <= UM_ID_FILE_MAX_AGE)
                    break;
            }
            if (expired(t0, timeout_seconds)) {
                um_set_error("um_comm_init_file: rank %d waited %d s for a fresh id record of this job (nonce %d) at %s", rank, timeout_seconds,
                             nonce, path);
                return UM_ERR_COLLECTIVE;
            }
            usleep(20000);
        }
        if (!touch(ack_path(rank))) {
            um_set_error("um_comm_init_file: rank %d cannot create %s.ack%d", rank, path, rank);
            return UM_ERR_COLLECTIVE;
        }
        while (!file_exists(go_path())) {
            if (expired(t0, timeout_seconds)) {
                um_set_error("um_comm_init_file: rank %d waited %d s for rank 0's go (another rank never acknowledged) at %s", rank,
                             timeout_seconds, path);
                (void)unlink(ack_path(rank));
                return UM_ERR_COLLECTIVE;
            }
            usleep(20000);
        }
    }
    const int e = um_comm_init_rank(comm_out, rec.id, rank, world);
    if (rank == 0) cleanup();                                       // every rank has joined (or the bootstrap failed): single use
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
