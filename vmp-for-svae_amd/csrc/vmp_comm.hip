// Multi-GPU exchange of the C ABI: one in-place all-reduce(sum) of the packed fp64 buffer over RCCL (xGMI inside a
// node).  Reference: the tower gather + M-step on the parameter device (experiments.py:247-260) and the gradient
// mean (helpers/tf_utils.py:52-87) - here every rank sums the same packed buffer and applies the identical update.
// RCCL is bound at run time with dlopen/dlsym: a torch process already carries an RCCL (soname librccl.so.1) and a
// second, link-time copy must not be pulled in next to it; single-GPU hosts never load it at all.
#include "vmp_common.h"
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>
#include <mutex>

using namespace vmp;

namespace {

struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_once;
char g_load_err[256] = "missing symbols";     // why the one-time load failed (written once, inside the call_once)

void load_rccl() {
    void* h = nullptr;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
        const char* e = dlerror();               // read ONCE: dlerror() clears the state it returns
        if (e) snprintf(g_load_err, sizeof(g_load_err), "%s", e);
    }
    if (!h) return;
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(h, "ncclAllReduce"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce;
}

int need_rccl(const char* who) {
    std::call_once(g_once, load_rccl);
    if (!g_rccl.ok) { set_error("%s: RCCL (librccl.so.1) could not be loaded: %s", who, g_load_err); return VMP_E_BADARG; }
    return 0;
}

int rccl_status(ncclResult_t r, const char* who) {
    if (r == ncclSuccess) return 0;
    set_error("%s: RCCL error %d (%s)", who, (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return (int)r > 0 ? (int)r : 1;
}

}  // namespace

extern "C" {

int vmp_comm_unique_id(void* id_out) {
    static_assert(sizeof(ncclUniqueId) == VMP_COMM_ID_BYTES, "RCCL unique id size");
    if (!id_out) { set_error("vmp_comm_unique_id: null pointer"); return VMP_E_BADARG; }
    if (int rc = need_rccl("vmp_comm_unique_id")) return rc;
    return rccl_status(g_rccl.GetUniqueId(static_cast<ncclUniqueId*>(id_out)), "vmp_comm_unique_id");
}

int vmp_comm_init_rank(void** comm_out, int nranks, const void* id, int rank) {
    if (!comm_out || !id || nranks < 1 || rank < 0 || rank >= nranks) { set_error("vmp_comm_init_rank: bad argument"); return VMP_E_BADARG; }
    if (int rc = need_rccl("vmp_comm_init_rank")) return rc;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    int rc = rccl_status(g_rccl.CommInitRank(&c, nranks, uid, rank), "vmp_comm_init_rank");
    if (rc == 0) *comm_out = c;               // untouched on failure
    return rc;
}

int vmp_comm_destroy(void* comm) {
    if (!comm) return 0;
    if (int rc = need_rccl("vmp_comm_destroy")) return rc;
    return rccl_status(g_rccl.CommDestroy(static_cast<ncclComm_t>(comm)), "vmp_comm_destroy");
}

int vmp_pack_allreduce(void* comm, double* buf, size_t n, void* stream) {
    if (!comm || (!buf && n)) { set_error("vmp_pack_allreduce: null pointer"); return VMP_E_BADARG; }
    if (n == 0) return 0;
    if (int rc = need_rccl("vmp_pack_allreduce")) return rc;
    return rccl_status(g_rccl.AllReduce(buf, buf, n, ncclFloat64, ncclSum, static_cast<ncclComm_t>(comm),
                                        static_cast<hipStream_t>(stream)), "vmp_pack_allreduce");
}

// ---- peer-visible exchange buffers (the one-launch distributed finalize of vmp_mix.hip) ------------------------------
// One buffer per rank, in device memory every peer can write while kernels run: uncached (MTYPE_UC) VRAM, exported with
// hipIpcGetMemHandle and mapped by the other ranks with hipIpcOpenMemHandle (xGMI peer access inside a node; two
// processes sharing one GPU map the same physical memory).  Layout: vmp_exch_bytes.
size_t vmp_exch_bytes(int nranks, int K, int D) {
    if (nranks < 1 || nranks > VMP_EXCH_MAX_RANKS || K < 1 || D < 1) return 0;
    const size_t swp = ((size_t)(2 + D + D * D) + 1 + 7) & ~(size_t)7;             // doubles per (sender, component) slot: moments + sum_k N_k
    return 2 * (size_t)nranks * K * swp * sizeof(double) + 2 * (size_t)nranks * K * sizeof(unsigned long long);
}

int vmp_exch_alloc(void** buf_out, size_t bytes) {
    if (!buf_out || !bytes) { set_error("vmp_exch_alloc: bad argument"); return VMP_E_BADARG; }
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("vmp_exch_alloc: %s", hipGetErrorString(e)); return (int)e; }
    e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(p); set_error("vmp_exch_alloc: %s", hipGetErrorString(e)); return (int)e; }
    *buf_out = p;
    return 0;
}

int vmp_exch_free(void* buf) {
    if (!buf) return 0;
    hipError_t e = hipFree(buf);
    if (e != hipSuccess) { set_error("vmp_exch_free: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

int vmp_exch_export(void* buf, void* handle_out) {
    static_assert(sizeof(hipIpcMemHandle_t) == VMP_EXCH_HANDLE_BYTES, "IPC handle size");
    if (!buf || !handle_out) { set_error("vmp_exch_export: null pointer"); return VMP_E_BADARG; }
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, buf);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("vmp_exch_export: %s", hipGetErrorString(e)); return (int)e; }
    memcpy(handle_out, &h, sizeof(h));
    return 0;
}

int vmp_exch_open(const void* handle, void** peer_out) {
    if (!handle || !peer_out) { set_error("vmp_exch_open: null pointer"); return VMP_E_BADARG; }
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("vmp_exch_open: %s", hipGetErrorString(e)); return (int)e; }
    *peer_out = p;
    return 0;
}

int vmp_exch_close(void* peer) {
    if (!peer) return 0;
    hipError_t e = hipIpcCloseMemHandle(peer);
    if (e != hipSuccess) { set_error("vmp_exch_close: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

}  // extern "C"
