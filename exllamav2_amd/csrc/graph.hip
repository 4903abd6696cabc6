// graph.hip -- HIP-graph capture / replay of a launch sequence (C ABI).
//
// Reference: Graph (graph.cu:20-164) captures each module's kernels after 205 eager calls and patches pointer / scalar
// kernel arguments node by node before every launch.  Here every per-step quantity (positions, sequence lengths, token
// ids) is read from device memory by the kernels themselves, so ONE captured graph covers a whole decode step
// (all layers + head + greedy sampling) and replays with no argument patching and no host round trip.
#include "hw.h"
#include "errors.h"

extern "C" {

int exl2_graph_begin_capture(void* stream)
{
    HIP_TRY(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed));
    return EXL2_OK;
}

int exl2_graph_end_capture(void* stream, void** graph_exec)
{
    EXL2_REQUIRE(graph_exec, "graph_end_capture: null output");
    hipGraph_t graph = nullptr;
    HIP_TRY(hipStreamEndCapture((hipStream_t)stream, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) EXL2_FAIL(EXL2_E_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    *graph_exec = exec;
    return EXL2_OK;
}

int exl2_graph_launch(void* graph_exec, void* stream)
{
    EXL2_REQUIRE(graph_exec, "graph_launch: null graph");
    HIP_TRY(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return EXL2_OK;
}

int exl2_graph_free(void* graph_exec)
{
    if (graph_exec) HIP_TRY(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return EXL2_OK;
}

}  // extern "C"
