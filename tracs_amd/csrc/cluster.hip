// cluster.hip -- threshold single-linkage clustering = connected components (gfx950).
//
// Reference behaviour restated: /root/reference/tracs/cluster.py:126-129 builds a CSR graph from
// the kept edges and calls scipy.sparse.csgraph.connected_components(directed=False).  SciPy
// numbers components in order of discovery from node 0 upwards, i.e. component id = rank of the
// component's smallest node id.  Here: lock-free union-find that always hooks the larger root
// under the smaller (so every component's root IS its smallest node, whatever the edge order),
// then an exclusive scan over "is root" gives the SciPy numbering.  Integer work, HBM/atomic bound.
#include "common.h"

namespace tracs {

// parent[] only ever moves a node towards smaller ids (hook larger root under smaller, path
// halving), so a stale read still lands on a valid ancestor: races cost retries, never a wrong
// component.  Agent-scope relaxed atomics keep the loads out of the (non-coherent) vector L1.
__device__ __forceinline__ int uf_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int uf_find(int *parent, int x)
{
    for (;;) {
        const int p = uf_load(&parent[x]);
        if (p == x) return x;
        const int gp = uf_load(&parent[p]);
        if (gp != p) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // path halving
        x = gp;
    }
}

__global__ void uf_init_kernel(int *__restrict__ parent, size_t n)
{
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (size_t)gridDim.x * blockDim.x) parent[v] = (int)v;
}

__global__ void uf_union_kernel(const int *__restrict__ I, const int *__restrict__ J, size_t n_edges, int *parent)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_edges; e += (size_t)gridDim.x * blockDim.x) {
        int a = I[e], b = J[e];
        for (;;) {
            a = uf_find(parent, a);
            b = uf_find(parent, b);
            if (a == b) break;
            if (a > b) { const int t = a; a = b; b = t; }     // a < b: hook root b under a
            const int old = atomicCAS(&parent[b], b, a);
            if (old == b) break;
            b = old;                                          // b was no longer a root: retry from its parent
        }
    }
}

// root[v] <- root of v (the forest is final: separate launch); flag[v] <- v is a root
__global__ void uf_roots_kernel(const int *__restrict__ parent, int *__restrict__ root, int *__restrict__ flag, size_t n)
{
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (size_t)gridDim.x * blockDim.x) {
        int x = (int)v;
        while (parent[x] != x) x = parent[x];
        root[v] = x;
        flag[v] = (x == (int)v) ? 1 : 0;
    }
}

// exclusive scan of int flags, single workgroup, chunked (n up to a few million nodes)
__global__ __launch_bounds__(1024) void scan_flags_kernel(int *__restrict__ flag, size_t n, int *__restrict__ total)
{
    __shared__ int part[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (size_t base = 0; base < n; base += 1024) {
        const size_t idx = base + threadIdx.x;
        const int v = idx < n ? flag[idx] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int t = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        const int incl = part[threadIdx.x];
        const int c0 = carry;
        __syncthreads();
        if (idx < n) flag[idx] = c0 + incl - v;
        if (threadIdx.x == 1023) carry = c0 + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ void uf_label_kernel(const int *__restrict__ root, const int *__restrict__ id_of, int *__restrict__ labels, size_t n)
{
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (size_t)gridDim.x * blockDim.x) labels[v] = id_of[root[v]];
}

struct CcWorkspaceIds { enum { PARENT = 16, FLAG, ROOT, TOTAL }; };

}  // namespace tracs

using namespace tracs;

extern "C" {

int tracs_connected_components_device(const int32_t *I, const int32_t *J, size_t n_edges, size_t n_nodes, int32_t *labels,
                                      int32_t *n_components_host, void *stream_)
{
    if (n_components_host) *n_components_host = 0;
    if (n_nodes == 0) return TRACS_OK;
    if (!labels || (n_edges && (!I || !J))) { set_error("tracs_connected_components_device: NULL argument"); return TRACS_E_ARG; }
    if (n_nodes >= (1ull << 31)) { set_error("connected_components: more than 2^31 nodes"); return TRACS_E_ARG; }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    DeviceCall guard(stream);
    int *parent, *flag, *root, *total;
    int rc;
    if ((rc = workspace_get(CcWorkspaceIds::PARENT, n_nodes * 4, reinterpret_cast<void **>(&parent)))) return rc;
    if ((rc = workspace_get(CcWorkspaceIds::FLAG, n_nodes * 4, reinterpret_cast<void **>(&flag)))) return rc;
    if ((rc = workspace_get(CcWorkspaceIds::ROOT, n_nodes * 4, reinterpret_cast<void **>(&root)))) return rc;
    if ((rc = workspace_get(CcWorkspaceIds::TOTAL, 64, reinterpret_cast<void **>(&total)))) return rc;
    const unsigned nb = (unsigned)std::min<size_t>((n_nodes + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(uf_init_kernel, dim3(nb), dim3(256), 0, stream, parent, n_nodes);
    if (n_edges) {
        const unsigned eb = (unsigned)std::min<size_t>((n_edges + 255) / 256, 256 * 16);
        hipLaunchKernelGGL(uf_union_kernel, dim3(eb), dim3(256), 0, stream, I, J, n_edges, parent);
    }
    hipLaunchKernelGGL(uf_roots_kernel, dim3(nb), dim3(256), 0, stream, parent, root, flag, n_nodes);
    hipLaunchKernelGGL(scan_flags_kernel, dim3(1), dim3(1024), 0, stream, flag, n_nodes, total);
    hipLaunchKernelGGL(uf_label_kernel, dim3(nb), dim3(256), 0, stream, root, flag, labels, n_nodes);
    TRACS_HIP_CHECK(hipGetLastError());
    if (n_components_host) {
        TRACS_HIP_CHECK(hipMemcpyAsync(n_components_host, total, 4, hipMemcpyDeviceToHost, stream));
        TRACS_HIP_CHECK(hipStreamSynchronize(stream));
    }
    return TRACS_OK;
}

int tracs_connected_components(const int32_t *I, const int32_t *J, size_t n_edges, size_t n_nodes, int32_t *labels,
                               int32_t *n_components)
{
    if (n_components) *n_components = 0;
    if (n_nodes == 0) return TRACS_OK;
    if (!labels || (n_edges && (!I || !J))) { set_error("tracs_connected_components: NULL argument"); return TRACS_E_ARG; }
    for (size_t e = 0; e < n_edges; e++)
        if (I[e] < 0 || J[e] < 0 || (size_t)I[e] >= n_nodes || (size_t)J[e] >= n_nodes) { set_error("connected_components: edge endpoint out of range"); return TRACS_E_ARG; }
    int *dI = nullptr, *dJ = nullptr, *dL = nullptr;
    auto cleanup = [&]() { if (dI) (void)hipFree(dI); if (dJ) (void)hipFree(dJ); if (dL) (void)hipFree(dL); };
#define CC_CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { cleanup(); set_error(std::string(#x ": ") + hipGetErrorString(e__)); return TRACS_E_HIP; } } while (0)
    if (n_edges) {
        CC_CHECK(hipMalloc(reinterpret_cast<void **>(&dI), n_edges * 4));
        CC_CHECK(hipMalloc(reinterpret_cast<void **>(&dJ), n_edges * 4));
        CC_CHECK(hipMemcpy(dI, I, n_edges * 4, hipMemcpyHostToDevice));
        CC_CHECK(hipMemcpy(dJ, J, n_edges * 4, hipMemcpyHostToDevice));
    }
    CC_CHECK(hipMalloc(reinterpret_cast<void **>(&dL), n_nodes * 4));
    int rc = tracs_connected_components_device(dI, dJ, n_edges, n_nodes, dL, n_components, nullptr);
    if (rc) { cleanup(); return rc; }
    CC_CHECK(hipMemcpy(labels, dL, n_nodes * 4, hipMemcpyDeviceToHost));
#undef CC_CHECK
    cleanup();
    return TRACS_OK;
}

}  // extern "C"
