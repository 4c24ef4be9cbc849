// Host-side runtime pieces of the C ABI: error string, hipGraph capture of a reverse step, HIP events.
#include "common.hpp"

namespace gsdd {
static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }
}  // namespace gsdd

using namespace gsdd;

extern "C" const char* gsdd_last_error(void) { return g_err.c_str(); }
extern "C" int gsdd_version(void) { return 101; }
extern "C" int64_t gsdd_abi_sizeof(int which) {
    switch (which) {
        case 0: return (int64_t)sizeof(gsdd_gemm_desc);
        case 1: return (int64_t)sizeof(gsdd_layer_desc);
        case 2: return (int64_t)sizeof(gsdd_step_desc);
        case 3: return (int64_t)sizeof(gsdd_train_desc);
        default: return -1;
    }
}

// ---------------------------------------------------------------- hipGraph capture
extern "C" int gsdd_graph_begin(void* stream) {
    GSDD_CHECK_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return GSDD_OK;
}

extern "C" int gsdd_graph_end(void* stream, void** graph_exec_out) {
    GSDD_CHECK_ARG(graph_exec_out != nullptr, "null output");
    hipGraph_t graph = nullptr;
    GSDD_CHECK_HIP(hipStreamEndCapture((hipStream_t)stream, &graph));
    if (graph == nullptr) {
        set_error("gsdd_graph_end: capture produced no graph");
        return GSDD_E_STATE;
    }
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    GSDD_CHECK_HIP(e);
    *graph_exec_out = (void*)exec;
    return GSDD_OK;
}

extern "C" int gsdd_graph_launch(void* graph_exec, void* stream) {
    GSDD_CHECK_ARG(graph_exec != nullptr, "null graph");
    GSDD_CHECK_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return GSDD_OK;
}

extern "C" int gsdd_graph_destroy(void* graph_exec) {
    if (graph_exec != nullptr) GSDD_CHECK_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return GSDD_OK;
}

// ---------------------------------------------------------------- events
extern "C" int gsdd_event_create(void** ev) {
    GSDD_CHECK_ARG(ev != nullptr, "null output");
    hipEvent_t e;
    GSDD_CHECK_HIP(hipEventCreate(&e));
    *ev = (void*)e;
    return GSDD_OK;
}
extern "C" int gsdd_event_record(void* ev, void* stream) {
    GSDD_CHECK_ARG(ev != nullptr, "null event");
    GSDD_CHECK_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return GSDD_OK;
}
extern "C" int gsdd_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms) {
    GSDD_CHECK_ARG(ev_start && ev_stop && ms, "null argument");
    GSDD_CHECK_HIP(hipEventSynchronize((hipEvent_t)ev_stop));
    GSDD_CHECK_HIP(hipEventElapsedTime(ms, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
    return GSDD_OK;
}
extern "C" int gsdd_event_destroy(void* ev) {
    if (ev != nullptr) GSDD_CHECK_HIP(hipEventDestroy((hipEvent_t)ev));
    return GSDD_OK;
}
