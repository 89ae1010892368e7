// rccl_sync.h — the barrier and the timing reduction of the CLI's process-per-GPU shape (-procs N / -world N -rank r) over RCCL (`-rccl 1`).
// north_star: "image-level parallelism shards pairs.txt across the 8 GPUs of one MI355X node with RCCL over xGMI only for work-stealing/barriers". Pairs are independent, so
// nothing of the data path crosses GPUs; what the ranks of a node do share is (a) a common start, (b) the job's wall time = the slowest rank's, (c) the number of pairs done —
// three all-reduces of one double each (SUM as the start barrier, MAX, SUM), exactly what bench.py's ranks do through torch.distributed (python/nct/shard.py: timed_region).
// Work stealing stays a lock-file ticket (one integer per pair; -steal 1): a collective is the wrong tool for an asynchronous counter.
// librccl and libamdhip64 are resolved with dlopen at run time, so the CLI builds with plain g++ and runs where RCCL is absent (then -rccl 1 reports why and the ranks go on
// without the barrier: a pair's result never depends on it). The unique id travels through <output>/.rccl_id (128 bytes + a run token; rank 0 writes it atomically).
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>

namespace rccl_sync {

struct UniqueId { char internal[128]; };                      // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value like the header's struct
enum { kFloat64 = 8, kSum = 0, kMax = 2 };                    // ncclFloat64, ncclSum, ncclMax (rccl.h)
enum { kH2D = 1, kD2H = 2 };                                  // hipMemcpyHostToDevice, hipMemcpyDeviceToHost

class Group {
public:
    // true = the communicator of `world` ranks exists; false = why() says what is missing (the caller goes on without)
    bool init(int world, int rank, int device, const std::string& id_path, uint64_t token, double timeout_s = 120.0) {
        world_ = world; rank_ = rank;
        hip_ = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!hip_) hip_ = dlopen("libamdhip64.so.7", RTLD_NOW | RTLD_GLOBAL);
        if (!hip_) return fail(std::string("libamdhip64.so: ") + dlerror());
        for (const char* n : {"librccl.so", "librccl.so.1"}) { rccl_ = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (rccl_) break; }
        if (!rccl_) return fail(std::string("librccl.so: ") + dlerror());
        if (!sym(hip_, "hipSetDevice", hipSetDevice_) || !sym(hip_, "hipMalloc", hipMalloc_) || !sym(hip_, "hipFree", hipFree_) || !sym(hip_, "hipMemcpy", hipMemcpy_) ||
            !sym(hip_, "hipDeviceSynchronize", hipDeviceSynchronize_) || !sym(rccl_, "ncclGetUniqueId", getid_) || !sym(rccl_, "ncclCommInitRank", initrank_) ||
            !sym(rccl_, "ncclAllReduce", allreduce_) || !sym(rccl_, "ncclCommDestroy", destroy_) || !sym(rccl_, "ncclGetErrorString", errstr_)) return false;
        if (hipSetDevice_(device) != 0) return fail("hipSetDevice(" + std::to_string(device) + ") failed");
        UniqueId id; memset(&id, 0, sizeof id);
        if (rank == 0) {
            const int rc = getid_(&id);
            if (rc != 0) return fail(std::string("ncclGetUniqueId: ") + errstr_(rc));
            const std::string tmp = id_path + ".tmp." + std::to_string((long)getpid());
            FILE* f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(&id, 1, sizeof id, f) != sizeof id || fwrite(&token, 1, sizeof token, f) != sizeof token || fclose(f) != 0) return fail("cannot write " + tmp);
            if (rename(tmp.c_str(), id_path.c_str()) != 0) return fail("cannot publish " + id_path);
        } else {
            const auto t0 = std::chrono::steady_clock::now();
            for (;;) {
                FILE* f = fopen(id_path.c_str(), "rb");
                uint64_t tk = ~token;
                const bool got = f && fread(&id, 1, sizeof id, f) == sizeof id && fread(&tk, 1, sizeof tk, f) == sizeof tk && tk == token;      // a file of another run carries another token
                if (f) fclose(f);
                if (got) break;
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return fail("rank 0's id did not appear in " + id_path);
                std::this_thread::sleep_for(std::chrono::milliseconds(20));
            }
        }
        const int rc = initrank_(&comm_, world, id, rank);
        if (rc != 0) { comm_ = nullptr; return fail(std::string("ncclCommInitRank: ") + errstr_(rc)); }
        if (hipMalloc_(&dbuf_, 2 * sizeof(double)) != 0) return fail("hipMalloc failed");
        ok_ = true;
        return true;
    }
    // all-reduce of one double over the ranks (device buffer, null stream); also a barrier: nobody returns before everybody has arrived
    bool reduce(double v, int op, double* out) {
        if (!ok_) return false;
        if (hipMemcpy_(dbuf_, &v, sizeof v, kH2D) != 0) return fail("hipMemcpy failed");
        const int rc = allreduce_(dbuf_, (char*)dbuf_ + sizeof(double), 1, kFloat64, op, comm_, nullptr);
        if (rc != 0) return fail(std::string("ncclAllReduce: ") + errstr_(rc));
        if (hipDeviceSynchronize_() != 0) return fail("hipDeviceSynchronize failed");
        if (hipMemcpy_(out, (char*)dbuf_ + sizeof(double), sizeof(double), kD2H) != 0) return fail("hipMemcpy failed");
        return true;
    }
    bool barrier() { double s = 0; return reduce(1.0, kSum, &s) && s == (double)world_; }
    void finish(const std::string& id_path) {
        if (dbuf_) { hipFree_(dbuf_); dbuf_ = nullptr; }
        if (comm_) { destroy_(comm_); comm_ = nullptr; }
        if (rank_ == 0) unlink(id_path.c_str());
        ok_ = false;
    }
    bool ok() const { return ok_; }
    const std::string& why() const { return err_; }

private:
    template <typename F> bool sym(void* h, const char* name, F& f) {
        f = reinterpret_cast<F>(dlsym(h, name));
        return f ? true : fail(std::string("symbol ") + name + " not found");
    }
    bool fail(const std::string& m) { err_ = m; ok_ = false; return false; }
    int world_ = 1, rank_ = 0;
    bool ok_ = false;
    std::string err_;
    void *hip_ = nullptr, *rccl_ = nullptr, *comm_ = nullptr, *dbuf_ = nullptr;
    int (*hipSetDevice_)(int) = nullptr;
    int (*hipMalloc_)(void**, size_t) = nullptr;
    int (*hipFree_)(void*) = nullptr;
    int (*hipMemcpy_)(void*, const void*, size_t, int) = nullptr;
    int (*hipDeviceSynchronize_)() = nullptr;
    int (*getid_)(UniqueId*) = nullptr;
    int (*initrank_)(void**, int, UniqueId, int) = nullptr;
    int (*allreduce_)(const void*, void*, size_t, int, int, void*, void*) = nullptr;
    int (*destroy_)(void*) = nullptr;
    const char* (*errstr_)(int) = nullptr;
};

}  // namespace rccl_sync
