// tests/emu/fake_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl.so.1 for the no-GPU container, so that the RCCL branch of
// csrc/ss_dist.hip (ss_comm_create_rccl: count matrices, offsets, grouped ncclSend / ncclRecv per peer, ncclAllGather / ncclAllReduce through
// staging buffers) runs between real PROCESSES -- tests/rccl_rank_worker.py under torch.distributed.run, each rank with the emulated
// library (tests/emu) -- and is compared with the single-context reconstruction bit for bit.  Built by tests/emu/build_emu.py --fake-rccl
// (libfake_rccl.so) and bound by the library through SPLASH_RCCL_LIB (ss_dist.hip, rccl_api).
//
// What it implements: the twelve entry points ss_dist.hip binds, for host-addressable buffers (the emulated "device" memory), messages as files in a
// directory named after the unique id (SS_FAKE_RCCL_DIR or /dev/shm).  Semantics it CHECKS: every receive finds a message from that peer, in
// order, of exactly the announced size and datatype; every call inside ncclGroupStart / ncclGroupEnd is issued at the closing ncclGroupEnd, sends
// before receives (a group cannot deadlock on its own order, as with NCCL).  What it cannot show: anything about the real library on real links.
#include <dirent.h>
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5,
               ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3, ncclAvg = 4 } ncclRedOp_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef void* hipStream_t;
}

struct ncclComm {
    int rank = 0, world = 1;
    std::string dir;
    std::vector<uint64_t> send_seq, recv_seq;  // per peer: messages sent to / received from it so far
    uint64_t coll_seq = 0;
};

namespace {
struct Pending { bool send; const void* sbuf; void* rbuf; size_t bytes; int dtype; int peer; ncclComm* comm; };
thread_local int g_depth = 0;
thread_local std::vector<Pending> g_pending;

size_t dtype_size(int t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}
double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}
std::string msg_path(ncclComm* c, const char* kind, int src, int dst, uint64_t seq) {
    char b[128];
    snprintf(b, sizeof(b), "/%s_%d_%d_%llu", kind, src, dst, (unsigned long long)seq);
    return c->dir + b;
}
ncclResult_t put(ncclComm* c, const std::string& path, const void* buf, size_t bytes, int dtype) {
    const std::string tmp = path + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) {
        fprintf(stderr, "[fake-rccl] rank %d: cannot create %s: %s\n", c->rank, tmp.c_str(), strerror(errno));
        return ncclSystemError;
    }
    const uint64_t head[2] = {(uint64_t)bytes, (uint64_t)dtype};
    bool ok = fwrite(head, sizeof(head), 1, f) == 1 && (bytes == 0 || fwrite(buf, 1, bytes, f) == bytes);
    ok = fclose(f) == 0 && ok;
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) return ncclSystemError;
    return ncclSuccess;
}
ncclResult_t get(ncclComm* c, const std::string& path, void* buf, size_t bytes, int dtype) {
    const char* e = getenv("SS_FAKE_RCCL_TIMEOUT_S");
    const double limit = e ? atof(e) : 120.0, t0 = now_s();
    FILE* f = nullptr;
    long spins = 0;
    while (!(f = fopen(path.c_str(), "rb"))) {
        if (now_s() - t0 > limit) {
            fprintf(stderr, "[fake-rccl] rank %d: no message %s after %.0f s (a receive without its send)\n", c->rank, path.c_str(), limit);
            return ncclSystemError;
        }
        usleep(++spins < 200 ? 50 : 1000);
    }
    uint64_t head[2] = {0, 0};
    bool ok = fread(head, sizeof(head), 1, f) == 1;
    if (ok && (head[0] != (uint64_t)bytes || head[1] != (uint64_t)dtype)) {
        fprintf(stderr, "[fake-rccl] rank %d: message %s carries %llu bytes of type %llu, the receive expects %zu bytes of type %d\n", c->rank, path.c_str(),
                (unsigned long long)head[0], (unsigned long long)head[1], bytes, dtype);
        fclose(f);
        return ncclInvalidArgument;
    }
    ok = ok && (bytes == 0 || fread(buf, 1, bytes, f) == bytes);
    fclose(f);
    unlink(path.c_str());
    return ok ? ncclSuccess : ncclSystemError;
}
ncclResult_t do_send(const Pending& p) {
    ncclComm* c = p.comm;
    return put(c, msg_path(c, "p2p", c->rank, p.peer, c->send_seq[p.peer]++), p.sbuf, p.bytes, p.dtype);
}
ncclResult_t do_recv(const Pending& p) {
    ncclComm* c = p.comm;
    return get(c, msg_path(c, "p2p", p.peer, c->rank, c->recv_seq[p.peer]++), p.rbuf, p.bytes, p.dtype);
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof(*id));
    unsigned long long a = (unsigned long long)getpid() ^ ((unsigned long long)time(nullptr) << 20), b = (unsigned long long)(now_s() * 1e9);
    snprintf(id->internal, sizeof(id->internal), "ssfr_%llx_%llx", a, b);
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int world, ncclUniqueId id, int rank) {
    if (!comm || world < 1 || rank < 0 || rank >= world) return ncclInvalidArgument;
    id.internal[sizeof(id.internal) - 1] = 0;
    if (strncmp(id.internal, "ssfr_", 5) != 0) return ncclInvalidArgument;
    ncclComm* c = new ncclComm();
    c->rank = rank;
    c->world = world;
    const char* base = getenv("SS_FAKE_RCCL_DIR");
    c->dir = std::string(base ? base : "/dev/shm") + "/" + id.internal;
    if (mkdir(c->dir.c_str(), 0700) != 0 && errno != EEXIST) {
        fprintf(stderr, "[fake-rccl] rank %d: mkdir %s: %s\n", rank, c->dir.c_str(), strerror(errno));
        delete c;
        return ncclSystemError;
    }
    c->send_seq.assign((size_t)world, 0);
    c->recv_seq.assign((size_t)world, 0);
    *comm = c;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    rmdir(c->dir.c_str());  // (succeeds for the last rank out, once every message has been consumed)
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t c) { return ncclCommDestroy(c); }
ncclResult_t ncclCommGetAsyncError(ncclComm_t, ncclResult_t* r) {
    if (r) *r = ncclSuccess;
    return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclSystemError: return "unhandled system error (fake RCCL: missing message or file error)";
        case ncclInvalidArgument: return "invalid argument (fake RCCL: size or type of a message does not match its receive)";
        case ncclInvalidUsage: return "invalid usage";
        default: return "error";
    }
}
ncclResult_t ncclGroupStart() {
    ++g_depth;
    return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    std::vector<Pending> ops;
    ops.swap(g_pending);
    ncclResult_t worst = ncclSuccess;
    for (const Pending& p : ops)
        if (p.send) {
            const ncclResult_t r = do_send(p);
            if (r != ncclSuccess) worst = r;
        }
    for (const Pending& p : ops)
        if (!p.send) {
            const ncclResult_t r = do_recv(p);
            if (r != ncclSuccess) worst = r;
        }
    return worst;
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t) {
    if (!c || peer < 0 || peer >= c->world) return ncclInvalidArgument;
    const Pending p{true, buf, nullptr, count * dtype_size(t), (int)t, peer, c};
    if (g_depth > 0) {
        g_pending.push_back(p);
        return ncclSuccess;
    }
    return do_send(p);
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t) {
    if (!c || peer < 0 || peer >= c->world) return ncclInvalidArgument;
    const Pending p{false, nullptr, buf, count * dtype_size(t), (int)t, peer, c};
    if (g_depth > 0) {
        g_pending.push_back(p);
        return ncclSuccess;
    }
    return do_recv(p);
}
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t) {
    if (!c) return ncclInvalidArgument;
    const size_t bytes = count * dtype_size(t);
    const uint64_t seq = c->coll_seq++;
    // (the send buffer may alias the caller's slot of the receive buffer: copy out first)
    std::vector<char> mine((const char*)send, (const char*)send + bytes);
    for (int q = 0; q < c->world; ++q)
        if (q != c->rank) {
            const ncclResult_t r = put(c, msg_path(c, "ag", c->rank, q, seq), mine.data(), bytes, (int)t);
            if (r != ncclSuccess) return r;
        }
    memcpy((char*)recv + (size_t)c->rank * bytes, mine.data(), bytes);
    for (int q = 0; q < c->world; ++q)
        if (q != c->rank) {
            const ncclResult_t r = get(c, msg_path(c, "ag", q, c->rank, seq), (char*)recv + (size_t)q * bytes, bytes, (int)t);
            if (r != ncclSuccess) return r;
        }
    return ncclSuccess;
}
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t) {
    if (!c || t != ncclUint32 || op != ncclSum) return ncclInvalidArgument;  // (all the library asks for)
    const size_t bytes = count * 4;
    const uint64_t seq = c->coll_seq++;
    std::vector<uint32_t> mine((const uint32_t*)send, (const uint32_t*)send + count), other(count);
    for (int q = 0; q < c->world; ++q)
        if (q != c->rank) {
            const ncclResult_t r = put(c, msg_path(c, "ar", c->rank, q, seq), mine.data(), bytes, (int)t);
            if (r != ncclSuccess) return r;
        }
    std::vector<uint32_t> sum = mine;
    for (int q = 0; q < c->world; ++q)
        if (q != c->rank) {
            const ncclResult_t r = get(c, msg_path(c, "ar", q, c->rank, seq), other.data(), bytes, (int)t);
            if (r != ncclSuccess) return r;
            for (size_t i = 0; i < count; ++i) sum[i] += other[i];
        }
    memcpy(recv, sum.data(), bytes);
    return ncclSuccess;
}
}
