// replay.hip -- a captured step that is a short chain of kernels, launched as kernels.
//
// Reference path replaced: none of the reference's own -- its SVI.step (pyro/infer/svi.py:134-162) re-runs the
// model every step.  This package replays a captured hipGraph instead; for the headline step that graph holds
// TWO kernel nodes.  A plan copies the nodes' launch parameters out of the graph (which stays alive and owns the
// argument blocks) and launches the kernels one after the other into the caller's stream.
// MEASURED (tools/graph_launch_host_cost.py, ROCm 7.2, config 2): either form costs the host ~10 us per enqueue.  A
// step that waits for its loss is FASTER through hipGraphLaunch (72.2 us against 75.7: the graph's packets are
// encoded ahead of time and reach the device sooner); with replays queued ahead of the host plain launches follow
// each other more closely (62.5 against 66.5 us per step).  Hence opt-in on the host side
// (pyro_amd/kernels.py::DIRECT_REPLAY); the first version of this file claimed a gain for the waiting step that was
// an artefact of the measuring script (replays enqueued without reading their loss let the host run ahead).
// Only graphs that are ONE chain of kernel nodes launched from host functions qualify; everything else stays a
// hipGraphLaunch.
#include "common.h"

#include <vector>

namespace pa {

struct DirectNode {
  void* func;
  dim3 grid, block;
  unsigned shared;
  void** params;
};
struct DirectPlan { std::vector<DirectNode> nodes; };

}  // namespace pa

extern "C" {

int pa_graph_direct_plan(void* hip_graph, int max_nodes, void** plan_out, int* n_nodes_out) {
  PA_REQUIRE(hip_graph && plan_out, "graph_direct_plan: NULL pointer");
  *plan_out = nullptr;
  if (n_nodes_out) *n_nodes_out = 0;
  hipGraph_t g = (hipGraph_t)hip_graph;
  size_t n = 0;
  hipError_t e = hipGraphGetNodes(g, nullptr, &n);
  if (e != hipSuccess) return pa::fail(PA_ERR_LAUNCH, "graph_direct_plan: hipGraphGetNodes: %s", hipGetErrorString(e));
  if (n_nodes_out) *n_nodes_out = (int)n;
  if (n == 0 || (int64_t)n > (int64_t)max_nodes) return PA_OK;            // (no plan: the caller keeps the graph)
  std::vector<hipGraphNode_t> nodes(n);
  e = hipGraphGetNodes(g, nodes.data(), &n);
  if (e != hipSuccess) return pa::fail(PA_ERR_LAUNCH, "graph_direct_plan: hipGraphGetNodes: %s", hipGetErrorString(e));
  size_t ne = 0;
  e = hipGraphGetEdges(g, nullptr, nullptr, &ne);
  if (e != hipSuccess) return pa::fail(PA_ERR_LAUNCH, "graph_direct_plan: hipGraphGetEdges: %s", hipGetErrorString(e));
  if (ne != n - 1) return PA_OK;                                          // not one chain
  std::vector<hipGraphNode_t> from(ne), to(ne);
  if (ne) {
    e = hipGraphGetEdges(g, from.data(), to.data(), &ne);
    if (e != hipSuccess) return pa::fail(PA_ERR_LAUNCH, "graph_direct_plan: hipGraphGetEdges: %s", hipGetErrorString(e));
  }
  // the chain's order: the node nobody points at, then its successors
  std::vector<int> next(n, -1), indeg(n, 0);
  auto index_of = [&](hipGraphNode_t x) {
    for (size_t i = 0; i < n; ++i) if (nodes[i] == x) return (int)i;
    return -1;
  };
  for (size_t k = 0; k < ne; ++k) {
    const int a = index_of(from[k]), b = index_of(to[k]);
    if (a < 0 || b < 0 || next[a] >= 0 || indeg[b] > 0) return PA_OK;     // a fork or a join
    next[a] = b;
    indeg[b] = 1;
  }
  int cur = -1;
  for (size_t i = 0; i < n; ++i)
    if (indeg[i] == 0) { if (cur >= 0) return PA_OK; cur = (int)i; }
  if (cur < 0) return PA_OK;
  pa::DirectPlan* plan = new pa::DirectPlan{};
  for (size_t step = 0; step < n; ++step, cur = next[cur]) {
    if (cur < 0) { delete plan; return PA_OK; }
    hipGraphNodeType ty;
    if (hipGraphNodeGetType(nodes[cur], &ty) != hipSuccess || ty != hipGraphNodeTypeKernel) { delete plan; return PA_OK; }
    hipKernelNodeParams p{};
    if (hipGraphKernelNodeGetParams(nodes[cur], &p) != hipSuccess || p.func == nullptr || p.kernelParams == nullptr ||
        p.extra != nullptr) { delete plan; return PA_OK; }
    // (a kernel launched from a host function: its attributes resolve; a module function's handle does not)
    hipFuncAttributes attr{};
    if (hipFuncGetAttributes(&attr, p.func) != hipSuccess) { (void)hipGetLastError(); delete plan; return PA_OK; }
    plan->nodes.push_back(pa::DirectNode{p.func, p.gridDim, p.blockDim, p.sharedMemBytes, p.kernelParams});
  }
  *plan_out = (void*)plan;
  return PA_OK;
}

int pa_graph_direct_launch(void* plan, pa_stream_t stream) {
  PA_REQUIRE(plan != nullptr, "graph_direct_launch: NULL plan");
  hipStream_t s = pa::as_stream(stream);
  for (const pa::DirectNode& nd : ((pa::DirectPlan*)plan)->nodes) {
    hipError_t e = hipLaunchKernel(nd.func, nd.grid, nd.block, nd.params, nd.shared, s);
    if (e != hipSuccess) return pa::fail(PA_ERR_LAUNCH, "graph_direct_launch: %s", hipGetErrorString(e));
  }
  return PA_OK;
}

int pa_graph_direct_free(void* plan) {
  delete (pa::DirectPlan*)plan;
  return PA_OK;
}

}  // extern "C"
