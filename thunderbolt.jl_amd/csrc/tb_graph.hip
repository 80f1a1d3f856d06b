// tb_graph.hip — HIP graphs behind the boundary (round 5): a sequence of enqueue-only tb_* calls captured once and replayed with one launch.
//
// Why: a time loop on a thin slab (the 27-layer share of one of eight GPUs: 0.41 ms of kernels per step, 0.095 ms per CG iteration) pays more for
// its host-side launches than a thick one — 57 µs of a 0.466 ms step, a third of a 0.136 ms CG iteration (profiles/r04_v3/slab27_kernel_stats.txt).
// The reference has no counterpart (its loops are host loops, src/solver/time/euler.jl:71-101); the ABI sequence a host replays is unchanged.
//
// Time: scalar arguments are frozen into a captured launch, so the forms and ionic models that read the time take it from a two-double slot on the
// device while a capture is open (tb_device::d_tslot = {t, cos 2πt}); the graph starts with a one-thread kernel that writes the slot, and
// tb_graph_launch(graph, t) re-parameterises exactly that node before it launches.
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "tb_internal.h"

namespace tb {
__global__ void k_set_time(double t, double ct, double *__restrict__ slot) { slot[0] = t; slot[1] = ct; }
} // namespace tb

struct tb_graph {
    tb_device *dev = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipGraphNode_t tnode = nullptr;
    // argument block of the time node (the node keeps pointers to these)
    double t = 0.0, ct = 1.0;
    double *slot = nullptr;
    void *args[3] = {nullptr, nullptr, nullptr};
    hipKernelNodeParams kp{};
    int n_nodes = 0;
    bool timed = false; // a captured kernel reads the time slot: the graph starts with the node that writes it
};

using namespace tb;

extern "C" {

int tb_graph_begin(tb_device *dev)
{
    TB_REQUIRE(dev, "tb_graph_begin: dev is NULL");
    TB_REQUIRE(!dev->capturing, "tb_graph_begin: a capture is already open on this device");
    if (!dev->stream) { // the legacy default stream cannot be captured (and trying leaves the runtime refusing every later call)
        set_error("tb_graph_begin: the device runs on the legacy default stream; put it on a stream of its own (tb_device_set_stream) before capturing");
        return TB_ERR_UNSUPPORTED;
    }
    TB_HIP(hipSetDevice(dev->id));
    if (!dev->d_tslot) {
        TB_HIP(hipMalloc((void **)&dev->d_tslot, 2 * sizeof(double)));
        const double init[2] = {0.0, 1.0};
        TB_HIP(hipMemcpy(dev->d_tslot, init, sizeof init, hipMemcpyHostToDevice));
    }
    TB_HIP(hipStreamSynchronize(dev->stream)); // whatever was enqueued before is not part of the graph
    dev->defer_before_capture = dev->defer_status;
    dev->defer_status = true;                  // status reads synchronise: not allowed inside a capture (tb_device_poll_status after a launch)
    hipError_t e = hipStreamBeginCapture(dev->stream, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) {
        dev->defer_status = dev->defer_before_capture;
        set_error("tb_graph_begin: hipStreamBeginCapture: %s", hipGetErrorString(e));
        return TB_ERR_HIP;
    }
    dev->capturing = true;
    dev->tslot_used = false;
    return TB_OK;
}

int tb_graph_end(tb_device *dev, tb_graph **out)
{
    TB_REQUIRE(dev && out, "tb_graph_end: NULL argument");
    *out = nullptr;
    TB_REQUIRE(dev->capturing, "tb_graph_end: no capture is open on this device");
    dev->capturing = false;
    dev->defer_status = dev->defer_before_capture;
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(dev->stream, &g);
    if (e != hipSuccess || !g) {
        (void)hipGetLastError();
        set_error("tb_graph_end: hipStreamEndCapture: %s (a call inside the capture synchronised, or failed)", hipGetErrorString(e));
        // HIP 7.0 keeps refusing work on a stream whose capture was invalidated ("operation failed due to a previous error during capture"), also after
        // hipStreamEndCapture: the library replaces a stream of its own; a stream handed in by the host (tb_device_set_stream) is the host's to replace
        if (dev->own_stream) {
            (void)hipStreamDestroy(dev->stream);
            dev->stream = nullptr;
            if (hipStreamCreateWithFlags(&dev->stream, hipStreamNonBlocking) != hipSuccess) { dev->stream = nullptr; dev->own_stream = false; }
            (void)hipGetLastError();
        }
        return TB_ERR_HIP;
    }
    auto gr = new tb_graph();
    gr->dev = dev; gr->graph = g; gr->slot = dev->d_tslot;
    auto fail = [&](const char *what, hipError_t err) {
        set_error("tb_graph_end: %s: %s", what, hipGetErrorString(err));
        (void)hipGetLastError();
        if (gr->exec) (void)hipGraphExecDestroy(gr->exec);
        (void)hipGraphDestroy(g);
        delete gr;
        return TB_ERR_UNSUPPORTED;
    };
    size_t nroots = 0, nnodes = 0;
    if ((e = hipGraphGetNodes(g, nullptr, &nnodes)) != hipSuccess) return fail("hipGraphGetNodes", e);
    gr->n_nodes = (int)nnodes;
    if ((e = hipGraphGetRootNodes(g, nullptr, &nroots)) != hipSuccess) return fail("hipGraphGetRootNodes", e);
    std::vector<hipGraphNode_t> roots(nroots);
    if (nroots && (e = hipGraphGetRootNodes(g, roots.data(), &nroots)) != hipSuccess) return fail("hipGraphGetRootNodes", e);
    // the time node in front of every root of the captured graph — only if a captured call handed the slot to a kernel (a CG iteration has no time)
    gr->timed = dev->tslot_used;
    if (!gr->timed) {
        if ((e = hipGraphInstantiate(&gr->exec, g, nullptr, nullptr, 0)) != hipSuccess) return fail("hipGraphInstantiate", e);
        *out = gr;
        return TB_OK;
    }
    gr->args[0] = &gr->t; gr->args[1] = &gr->ct; gr->args[2] = &gr->slot;
    gr->kp.func = (void *)k_set_time;
    gr->kp.gridDim = dim3(1); gr->kp.blockDim = dim3(1); gr->kp.sharedMemBytes = 0;
    gr->kp.kernelParams = gr->args; gr->kp.extra = nullptr;
    if ((e = hipGraphAddKernelNode(&gr->tnode, g, nullptr, 0, &gr->kp)) != hipSuccess) return fail("hipGraphAddKernelNode", e);
    if (nroots) {
        std::vector<hipGraphNode_t> from(nroots, gr->tnode);
        if ((e = hipGraphAddDependencies(g, from.data(), roots.data(), nroots)) != hipSuccess) return fail("hipGraphAddDependencies", e);
    }
    if ((e = hipGraphInstantiate(&gr->exec, g, nullptr, nullptr, 0)) != hipSuccess) return fail("hipGraphInstantiate", e);
    *out = gr;
    return TB_OK;
}

int tb_graph_launch(tb_graph *g, double t)
{
    TB_REQUIRE(g && g->exec, "tb_graph_launch: NULL graph");
    TB_HIP(hipSetDevice(g->dev->id));
    if (g->timed && t != g->t) {
        g->t = t; g->ct = std::cos(2.0 * 3.141592653589793 * t);
        hipError_t e = hipGraphExecKernelNodeSetParams(g->exec, g->tnode, &g->kp);
        if (e != hipSuccess) { set_error("tb_graph_launch: hipGraphExecKernelNodeSetParams: %s", hipGetErrorString(e)); (void)hipGetLastError(); return TB_ERR_UNSUPPORTED; }
    }
    TB_HIP(hipGraphLaunch(g->exec, g->dev->stream));
    return TB_OK;
}

int tb_graph_node_count(tb_graph *g, int *n)
{
    TB_REQUIRE(g && n, "tb_graph_node_count: NULL argument");
    *n = g->n_nodes;
    return TB_OK;
}

int tb_graph_destroy(tb_graph *g)
{
    if (!g) return TB_OK;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return TB_OK;
}

} // extern "C"
