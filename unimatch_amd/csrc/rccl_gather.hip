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
#include <dirent.h>
#include <dlfcn.h>
#include <limits.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <algorithm>
#include <mutex>
#include <string>
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
//   1. publish   rank 0 removes whatever is at `path` (and every `<path>.ack* / .go* / .abort*` file an earlier job left), writes
//                {magic, world, nonce, stamp, DEADLINE, TOKEN, id} to `<path>.tmp` and renames it to `path` (atomic on one filesystem).
//                TOKEN is a fresh random 64-bit value per publish; every auxiliary file of this rendezvous carries it in its name,
//                so files of a crashed job under the same path and nonce can never be mistaken for this job's (round-5 ADVICE: a
//                stale `.go` sent a reader of the relaunch into ncclCommInitRank with a dead id).
//   2. ack       every other rank polls for a record with the right magic, world and NONCE whose deadline has not passed, then
//                creates `<path>.ack<rank>.<token>`; rank 0 polls for the world - 1 acks.  A reader keeps following the record: when it
//                is replaced by a new publish (the one it had found was a crashed job's leftover) it withdraws its ack and
//                acknowledges the new record; and it only accepts a go that is not older than its own ack.
//   3. go        rank 0 creates `<path>.go.<token>` -- only while (a) every ack STILL exists, (b) no `<path>.abort<rank>.<token>`
//                exists and (c) the shared deadline is more than a margin away; the others poll for it; everybody calls
//                ncclCommInitRank; rank 0 removes the files.
// ONE deadline for all ranks: stamp + rank 0's timeout, published in the record (round-5 ADVICE: with per-rank clocks an early
// reader could expire and leave after rank 0 had counted its ack, and rank 0 released the rest into ncclCommInitRank without it).
// A reader gives up only AFTER the deadline (it leaves an abort marker and withdraws its ack); rank 0 gives the go only BEFORE
// deadline - margin and re-checks acks and abort markers at that moment: the two decisions cannot cross unless a process is stalled
// for longer than the margin between its clock read and its file operation.  Before a record is found a reader waits its own
// `timeout_seconds`.  A rank that is missing, or that disagrees about the nonce, makes all ranks return UM_ERR_COLLECTIVE.
// Two jobs must not share a path at the same time: key it by job.  The nonce must be derived from values EVERY rank of the job
// shares (the path itself, MASTER_ADDR:MASTER_PORT, an explicit UM_RCCL_NONCE) -- never from a pid (unimatch_amd.dist.job_nonce).
#define UM_ID_FILE_MAX_AGE 600
namespace {
struct IdRecord {
    char magic[8];
    int world;
    int nonce;                  // per-job tag chosen by the caller (0: um_comm_init_file)
    long long stamp;            // wall clock of the publish, milliseconds
    long long deadline;         // wall clock (ms) after which every rank of this rendezvous gives up (LLONG_MAX: never)
    unsigned long long token;   // random per publish: part of every auxiliary file's name
    unsigned char id[UM_COMM_ID_BYTES];
};
const char kIdMagic[8] = {'U', 'M', 'R', 'C', 'C', 'L', '0', '4'};      // 04: shared deadline + per-publish token (03: ack / go)

bool file_exists(const char* p) { return access(p, F_OK) == 0; }
bool touch(const char* p) {
    FILE* f = fopen(p, "wb");
    if (!f) return false;
    fclose(f);
    return true;
}
long long now_ms() {
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    return (long long)ts.tv_sec * 1000 + ts.tv_nsec / 1000000;
}
unsigned long long fresh_token() {
    unsigned long long t = 0;
    FILE* f = fopen("/dev/urandom", "rb");
    if (f) {
        if (fread(&t, 1, sizeof(t), f) != sizeof(t)) t = 0;
        fclose(f);
    }
    if (t == 0) t = ((unsigned long long)now_ms() * 6364136223846793005ull) ^ ((unsigned long long)getpid() << 32) ^ 0x9e3779b97f4a7c15ull;
    return t;
}
// unlink every `<path>.ack*`, `<path>.go*`, `<path>.abort*` (whatever token they carry)
void remove_aux_files(const char* path) {
    std::string p(path);
    const size_t slash = p.rfind('/');
    const std::string dir = slash == std::string::npos ? "." : (slash == 0 ? "/" : p.substr(0, slash));
    const std::string base = slash == std::string::npos ? p : p.substr(slash + 1);
    DIR* d = opendir(dir.c_str());
    if (!d) return;
    while (struct dirent* e = readdir(d)) {
        const std::string name(e->d_name);
        if (name.size() <= base.size() || name.compare(0, base.size(), base) != 0) continue;
        const std::string rest = name.substr(base.size());
        if (rest.compare(0, 4, ".ack") == 0 || rest.compare(0, 3, ".go") == 0 || rest.compare(0, 6, ".abort") == 0)
            (void)unlink((dir + "/" + name).c_str());
    }
    closedir(d);
}
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
    char aux[1024 + 64];
    auto ack_path = [&](int r) { snprintf(aux, sizeof(aux), "%s.ack%d.%016llx", path, r, rec.token); return aux; };
    auto abort_path = [&](int r) { snprintf(aux, sizeof(aux), "%s.abort%d.%016llx", path, r, rec.token); return aux; };
    auto go_path = [&]() { snprintf(aux, sizeof(aux), "%s.go.%016llx", path, rec.token); return aux; };
    auto cleanup = [&]() {                                          // rank 0 only
        (void)unlink(path);
        remove_aux_files(path);
    };
    const long long t0 = now_ms();
    const long long own_deadline = timeout_seconds >= 0 ? t0 + 1000LL * timeout_seconds : LLONG_MAX;
    // rank 0 gives the go only while the deadline is further away than this (a reader gives up only after the deadline)
    const long long margin = timeout_seconds >= 0 ? std::min(5000LL, std::max(250LL, 250LL * timeout_seconds)) : 0;
    if (rank == 0) {
        memset(&rec, 0, sizeof(rec));
        memcpy(rec.magic, kIdMagic, sizeof(kIdMagic));
        rec.world = world;
        rec.nonce = nonce;
        rec.stamp = t0;
        rec.deadline = own_deadline;
        rec.token = fresh_token();
        if (int e = um_comm_unique_id(rec.id)) return e;
        cleanup();                                                  // leftovers of an earlier job under the same key
        char tmp[1024 + 8];
        snprintf(tmp, sizeof(tmp), "%s.tmp", path);
        FILE* f = fopen(tmp, "wb");
        if (!f || fwrite(&rec, 1, sizeof(rec), f) != sizeof(rec) || fclose(f) != 0 || rename(tmp, path) != 0) {
            um_set_error("um_comm_init_file: cannot publish the unique id at %s", path);
            return UM_ERR_COLLECTIVE;
        }
        for (;;) {
            int missing = 0, aborted = 0;
            for (int r = 1; r < world; ++r) {
                if (file_exists(abort_path(r))) aborted = r;
                else if (!file_exists(ack_path(r))) missing = r;
            }
            const long long now = now_ms();
            if (aborted || now > rec.deadline - margin) {
                if (aborted)
                    um_set_error("um_comm_init_file: rank %d gave up on the rendezvous (nonce %d) at %s", aborted, nonce, path);
                else
                    um_set_error("um_comm_init_file: rank 0 waited %d s for rank %d to acknowledge the id record (nonce %d) at %s -- is that "
                                 "rank running, on the same filesystem, with the same nonce?", timeout_seconds, missing ? missing : 1, nonce, path);
                cleanup();
                return UM_ERR_COLLECTIVE;
            }
            if (!missing) break;                                   // every ack present, no abort marker, deadline > margin away: go
            usleep(20000);
        }
        if (!touch(go_path())) {
            um_set_error("um_comm_init_file: cannot create %s.go", path);
            cleanup();
            return UM_ERR_COLLECTIVE;
        }
    } else {
        // A reader follows the record at `path`: it acknowledges the first valid record of its job, and if the record is REPLACED
        // while it waits (a new publish = a new token: the record it had found was a crashed job's leftover and rank 0 of this job has
        // only just arrived) it withdraws that ack and acknowledges the new one.  A go counts only when it is not older than the
        // reader's own ack (rank 0 writes it after it has seen every ack): the go a crashed job left next to its record is older.
        bool have = false;
        struct timespec ack_time = {0, 0};
        for (;;) {
            IdRecord cur;
            bool ok = false;
            if (FILE* f = fopen(path, "rb")) {
                const size_t n = fread(&cur, 1, sizeof(cur), f);
                fclose(f);
                const long long now = now_ms();
                ok = n == sizeof(cur) && memcmp(cur.magic, kIdMagic, sizeof(kIdMagic)) == 0 && cur.world == world && cur.nonce == nonce &&
                     now <= cur.deadline && now - cur.stamp <= 1000LL * UM_ID_FILE_MAX_AGE;
            }
            if (ok && (!have || cur.token != rec.token)) {
                if (have) (void)unlink(ack_path(rank));             // (ack_path reads rec.token: still the old record's)
                rec = cur;
                have = true;
                struct stat st;
                if (!touch(ack_path(rank)) || stat(ack_path(rank), &st) != 0) {
                    um_set_error("um_comm_init_file: rank %d cannot create its acknowledgement next to %s", rank, path);
                    return UM_ERR_COLLECTIVE;
                }
                ack_time = st.st_mtim;
            }
            if (have) {
                struct stat st;
                if (stat(go_path(), &st) == 0 &&
                    (st.st_mtim.tv_sec > ack_time.tv_sec || (st.st_mtim.tv_sec == ack_time.tv_sec && st.st_mtim.tv_nsec >= ack_time.tv_nsec)))
                    break;
                if (now_ms() > rec.deadline) {                       // the SHARED deadline: rank 0 no longer gives the go after it
                    (void)touch(abort_path(rank));
                    (void)unlink(ack_path(rank));
                    um_set_error("um_comm_init_file: rank %d waited until the rendezvous' deadline for rank 0's go (another rank never "
                                 "acknowledged) at %s", rank, path);
                    return UM_ERR_COLLECTIVE;
                }
            } else if (now_ms() > own_deadline) {
                um_set_error("um_comm_init_file: rank %d waited %d s for a fresh id record of this job (nonce %d) at %s", rank, timeout_seconds,
                             nonce, path);
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
