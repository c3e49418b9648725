// K10: connected components of the (representative, member) / ortholog edge list = the PARTITION that the
// reference's union-find produces in get_gene_group (PEPPAN.py:1598-1607).  Lock-free union-find: every edge
// hooks the larger root under the smaller with an atomic compare-and-swap, so the final root of a component is
// its smallest node id whatever the execution order; a second kernel flattens.  HBM-bound: 8 B per edge read,
// 4 B per node written; parent look-ups are random 4-byte reads.
#include "common.h"
#include <cstring>

namespace {

__device__ __forceinline__ uint32_t uf_root(uint32_t *parent, uint32_t x)
{
    for (;;) {
        const uint32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == x) return x;
        x = p;
    }
}

__global__ __launch_bounds__(256) void uf_init(uint32_t *parent, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) parent[i] = i;
}

__global__ __launch_bounds__(256) void uf_union(uint32_t *parent, const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, uint64_t m)
{
    const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= m) return;
    uint32_t x = a[e], y = b[e];
    for (;;) {
        x = uf_root(parent, x);
        y = uf_root(parent, y);
        if (x == y) return;
        if (x < y) { const uint32_t t = x; x = y; y = t; }          // x = larger root, hooked under y
        if (atomicCAS(&parent[x], x, y) == x) return;
    }
}

// the same over the edges of a hit table that is still on the device: (hit.q + q_base, node_of_target[hit.t])
__global__ __launch_bounds__(256) void uf_union_hits(uint32_t *parent, const pep_hit *__restrict__ hits, uint64_t m, uint32_t q_base, const uint32_t *__restrict__ node_of_target,
                                                     const uint32_t *__restrict__ d_m = nullptr)
{
    if (d_m) m = *d_m;                       // (the grid is sized from an upper bound then)
    const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= m) return;
    uint32_t x = hits[e].q + q_base, y = node_of_target[hits[e].t];
    for (;;) {
        x = uf_root(parent, x);
        y = uf_root(parent, y);
        if (x == y) return;
        if (x < y) { const uint32_t t = x; x = y; y = t; }
        if (atomicCAS(&parent[x], x, y) == x) return;
    }
}

__global__ __launch_bounds__(256) void uf_flatten(uint32_t *parent, uint32_t *__restrict__ label, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) label[i] = uf_root(parent, i);
}

}  // namespace

int pep_k10_components(pep_ctx *ctx, uint32_t n_nodes, uint64_t n_edges, const uint32_t *h_a, const uint32_t *h_b, uint32_t *h_label)
{
    if (n_nodes == 0) return PEP_OK;
    for (uint64_t e = 0; e < n_edges; ++e)
        if (h_a[e] >= n_nodes || h_b[e] >= n_nodes) return pep_fail(ctx, PEP_ERR_ARG, "pep_components: edge endpoint out of range");
    PEP_TRY(dev_reserve(ctx, ctx->ws[0], (size_t)n_nodes * 4));
    PEP_TRY(dev_reserve(ctx, ctx->ws[1], (size_t)n_nodes * 4));
    PEP_TRY(dev_reserve(ctx, ctx->ws[2], (n_edges + 1) * 4));
    PEP_TRY(dev_reserve(ctx, ctx->ws[3], (n_edges + 1) * 4));
    uint32_t *parent = ctx->ws[0].as<uint32_t>(), *label = ctx->ws[1].as<uint32_t>();
    hipLaunchKernelGGL(uf_init, dim3((unsigned)ceil_div(n_nodes, 256)), dim3(256), 0, ctx->stream, parent, n_nodes);
    if (n_edges) {
        PEP_HIP(ctx, hipMemcpyAsync(ctx->ws[2].p, h_a, n_edges * 4, hipMemcpyHostToDevice, ctx->stream));
        PEP_HIP(ctx, hipMemcpyAsync(ctx->ws[3].p, h_b, n_edges * 4, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(uf_union, dim3((unsigned)ceil_div(n_edges, 256)), dim3(256), 0, ctx->stream, parent, ctx->ws[2].as<const uint32_t>(),
                           ctx->ws[3].as<const uint32_t>(), n_edges);
    }
    hipLaunchKernelGGL(uf_flatten, dim3((unsigned)ceil_div(n_nodes, 256)), dim3(256), 0, ctx->stream, parent, label, n_nodes);
    PEP_HIP(ctx, hipGetLastError());
    // through pinned memory (a copy into the caller's pageable array is staged by the runtime: ~20 us before it even starts)
    if (pin_reserve(ctx, ctx->pin_labels, (size_t)n_nodes * 4) == PEP_OK) {
        PEP_HIP(ctx, hipMemcpyAsync(ctx->pin_labels.p, label, (size_t)n_nodes * 4, hipMemcpyDeviceToHost, ctx->stream));
        PEP_HIP(ctx, pep_stream_wait(ctx));
        memcpy(h_label, ctx->pin_labels.p, (size_t)n_nodes * 4);
    } else {
        PEP_HIP(ctx, hipMemcpyAsync(h_label, label, (size_t)n_nodes * 4, hipMemcpyDeviceToHost, ctx->stream));
        PEP_HIP(ctx, pep_stream_wait(ctx));
    }
    return PEP_OK;
}

// K10 straight from the device copy of the newest search's hit table (d_hits: ws[23], still intact) - no edge columns built or uploaded.
// The node map is uploaded only when it differs from the one the context already holds.
int pep_k10_components_dev(pep_ctx *ctx, uint32_t n_nodes, uint64_t n_hits, const pep_hit *d_hits, uint32_t q_base, const uint32_t *h_node_of_target,
                           uint64_t n_targets, uint32_t *h_label)
{
    if (n_nodes == 0) return PEP_OK;
    if (ctx->uf_nodes_host.size() != n_targets || (n_targets && memcmp(ctx->uf_nodes_host.data(), h_node_of_target, n_targets * 4) != 0)) {
        for (uint64_t t = 0; t < n_targets; ++t)
            if (h_node_of_target[t] >= n_nodes) return pep_fail(ctx, PEP_ERR_ARG, "pep_components_of_result: node of a target out of range");
        ctx->uf_nodes_host.assign(h_node_of_target, h_node_of_target + n_targets);
        PEP_TRY(dev_reserve(ctx, ctx->uf_nodes, (n_targets + 1) * 4));
        if (n_targets) PEP_HIP(ctx, hipMemcpyAsync(ctx->uf_nodes.p, ctx->uf_nodes_host.data(), n_targets * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    PEP_TRY(dev_reserve(ctx, ctx->ws[0], (size_t)n_nodes * 4));
    PEP_TRY(dev_reserve(ctx, ctx->ws[1], (size_t)n_nodes * 4));
    uint32_t *parent = ctx->ws[0].as<uint32_t>(), *label = ctx->ws[1].as<uint32_t>();
    hipLaunchKernelGGL(uf_init, dim3((unsigned)ceil_div(n_nodes, 256)), dim3(256), 0, ctx->stream, parent, n_nodes);
    if (n_hits) hipLaunchKernelGGL(uf_union_hits, dim3((unsigned)ceil_div(n_hits, 256)), dim3(256), 0, ctx->stream, parent, d_hits, n_hits, q_base, ctx->uf_nodes.as<const uint32_t>());
    hipLaunchKernelGGL(uf_flatten, dim3((unsigned)ceil_div(n_nodes, 256)), dim3(256), 0, ctx->stream, parent, label, n_nodes);
    PEP_HIP(ctx, hipGetLastError());
    // through pinned memory (a copy into the caller's pageable array is staged by the runtime: ~20 us before it even starts)
    if (pin_reserve(ctx, ctx->pin_labels, (size_t)n_nodes * 4) == PEP_OK) {
        PEP_HIP(ctx, hipMemcpyAsync(ctx->pin_labels.p, label, (size_t)n_nodes * 4, hipMemcpyDeviceToHost, ctx->stream));
        PEP_HIP(ctx, pep_stream_wait(ctx));
        memcpy(h_label, ctx->pin_labels.p, (size_t)n_nodes * 4);
    } else {
        PEP_HIP(ctx, hipMemcpyAsync(h_label, label, (size_t)n_nodes * 4, hipMemcpyDeviceToHost, ctx->stream));
        PEP_HIP(ctx, pep_stream_wait(ctx));
    }
    return PEP_OK;
}

// K10 as the tail of a search (pep_set_grouping): queued behind `emit` on the search's stream, BEFORE the search's final synchronisation -
// no second call, no second wait, and uf_flatten writes the labels straight into pinned host memory (no blit behind it).  The labels are
// in ctx->pin_labels once the stream has been waited for.
int pep_k10_queue(pep_ctx *ctx, uint64_t n_hits, const pep_hit *d_hits, const uint32_t *d_n_hits)
{
    const uint32_t n_nodes = ctx->grp_nodes;
    if (n_nodes == 0) return PEP_OK;
    if (ctx->uf_nodes_host.size() < ctx->t.n) return pep_fail(ctx, PEP_ERR_STATE, "pep_set_grouping: fewer node entries than the search has targets");
    if ((uint64_t)ctx->q.n + ctx->grp_q_base > n_nodes) return pep_fail(ctx, PEP_ERR_STATE, "pep_set_grouping: query nodes beyond n_nodes");
    PEP_TRY(dev_reserve(ctx, ctx->ws[0], (size_t)n_nodes * 4));
    PEP_TRY(pin_reserve(ctx, ctx->pin_labels, (size_t)n_nodes * 4));
    uint32_t *parent = ctx->ws[0].as<uint32_t>();
    if (!(ctx->ext.pending && ctx->ext.parent_ready))          // (pack_out has initialised the parents on its way)
        hipLaunchKernelGGL(uf_init, dim3((unsigned)ceil_div(n_nodes, 256)), dim3(256), 0, ctx->stream, parent, n_nodes);
    if (n_hits) hipLaunchKernelGGL(uf_union_hits, dim3((unsigned)ceil_div(n_hits, 256)), dim3(256), 0, ctx->stream, parent, d_hits, n_hits, ctx->grp_q_base, ctx->uf_nodes.as<const uint32_t>(), d_n_hits);
    hipLaunchKernelGGL(uf_flatten, dim3((unsigned)ceil_div(n_nodes, 256)), dim3(256), 0, ctx->stream, parent, reinterpret_cast<uint32_t *>(ctx->pin_labels.p), n_nodes);
    PEP_HIP(ctx, hipGetLastError());
    return PEP_OK;
}

int pep_k10_set_grouping(pep_ctx *ctx, uint32_t n_nodes, uint32_t q_base, const uint32_t *h_node_of_target, uint64_t n_targets)
{
    ctx->grp_nodes = 0;
    if (n_nodes == 0) return PEP_OK;
    if (ctx->uf_nodes_host.size() != n_targets || (n_targets && memcmp(ctx->uf_nodes_host.data(), h_node_of_target, n_targets * 4) != 0)) {
        for (uint64_t t = 0; t < n_targets; ++t)
            if (h_node_of_target[t] >= n_nodes) return pep_fail(ctx, PEP_ERR_ARG, "pep_set_grouping: node of a target out of range");
        ctx->uf_nodes_host.assign(h_node_of_target, h_node_of_target + n_targets);
        PEP_TRY(dev_reserve(ctx, ctx->uf_nodes, (n_targets + 1) * 4));
        if (n_targets) PEP_HIP(ctx, hipMemcpy(ctx->uf_nodes.p, ctx->uf_nodes_host.data(), n_targets * 4, hipMemcpyHostToDevice));
    } else {
        for (uint64_t t = 0; t < n_targets; ++t)
            if (h_node_of_target[t] >= n_nodes) return pep_fail(ctx, PEP_ERR_ARG, "pep_set_grouping: node of a target out of range");
    }
    ctx->grp_nodes = n_nodes;
    ctx->grp_q_base = q_base;
    return PEP_OK;
}
