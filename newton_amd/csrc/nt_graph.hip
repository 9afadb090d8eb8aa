// nt_graph.hip -- hipGraph capture of a caller's frame through the C ABI (include/newton_hip.h: nt_graph_*).
//
// Reference behaviour: Newton's examples record simulate() -- clear_forces / collide / step per substep -- with wp.ScopedCapture and
// replay it with wp.capture_launch (newton/examples/basic/example_basic_urdf.py:112-141).  Every entry point of this library launches
// on the caller's stream, owns no memory and never reads back, so the same frame records into one hipGraph; these four calls are
// the capture / replay helper for hosts that do not bring their own (the Python package can also use torch's).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/newton_hip.h"

struct nt_graph {
    hipGraph_t graph;
    hipGraphExec_t exec;
};

extern "C" {

#ifdef NT_EMULATED_GRID  // (tests/emu runs kernels synchronously on host threads: there is nothing to capture)
nt_status nt_graph_capture_begin(void*) { return NT_ERR_UNSUPPORTED; }
nt_status nt_graph_capture_end(void*, nt_graph**) { return NT_ERR_UNSUPPORTED; }
nt_status nt_graph_launch(nt_graph*, void*) { return NT_ERR_UNSUPPORTED; }
void nt_graph_destroy(nt_graph*) {}
#else
nt_status nt_graph_capture_begin(void* stream) {
    if (!stream) return NT_ERR_INVALID_ARG;  // the legacy default stream cannot be captured
    return hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal) == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_graph_capture_end(void* stream, nt_graph** out) {
    if (!stream || !out) return NT_ERR_INVALID_ARG;
    *out = nullptr;
    hipGraph_t g = nullptr;
    if (hipStreamEndCapture((hipStream_t)stream, &g) != hipSuccess || !g) return NT_ERR_LAUNCH;
    hipGraphExec_t e = nullptr;
    if (hipGraphInstantiate(&e, g, nullptr, nullptr, 0) != hipSuccess || !e) {
        hipGraphDestroy(g);
        return NT_ERR_LAUNCH;
    }
    *out = new nt_graph{g, e};
    return NT_OK;
}

nt_status nt_graph_launch(nt_graph* g, void* stream) {
    if (!g || !g->exec) return NT_ERR_INVALID_ARG;
    return hipGraphLaunch(g->exec, (hipStream_t)stream) == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

void nt_graph_destroy(nt_graph* g) {
    if (!g) return;
    if (g->exec) hipGraphExecDestroy(g->exec);
    if (g->graph) hipGraphDestroy(g->graph);
    delete g;
}
#endif

}  // extern "C"
