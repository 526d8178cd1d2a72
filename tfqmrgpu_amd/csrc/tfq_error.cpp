// Status decoding: tfqmrgpuGetErrorString / tfqmrgpuPrintError.
// A status is code + 1000*line + 10^7*character (include/tfqmrgpu.h); the messages say the same
// things as the reference decoder (real-space/tfQMRgpu tfQMRgpu/source/tfqmrgpu_error_tool.cxx:33-76)
// so that logs of existing applications keep their meaning.  Like the reference, the string lives in
// one static buffer (not thread-safe by contract).
#include <cstdio>
#include <cstring>

#include "tfqmrgpu.h"

namespace {
struct Entry { int code; char const* text; int args; }; // args: 0 none, 1 line, 2 (key as int, line), 3 (key as char, line)
Entry const kTable[] = {
    {TFQMRGPU_STATUS_MAX_ITERATIONS,    "tfQMRgpu: Max number of iterations exceeded!", 0},
    {TFQMRGPU_STATUS_BREAKDOWN,         "tfQMRgpu: All components have broken down!", 0},
    {TFQMRGPU_STATUS_NO_INFO_PASSED,    "tfQMRgpu: getInfo did not provide any information!", 0},
    {TFQMRGPU_POINTER_INVALID,          "tfQMRgpu: Pointer invalid at line %d!", 1},
    {TFQMRGPU_STATUS_ALLOCATION_FAILED, "tfQMRgpu: Allocation failed at line %d!", 1},
    {TFQMRGPU_STATUS_RANDOM_GEN_FAILED, "tfQMRgpu: Random number generation line %d!", 1},
    {TFQMRGPU_NO_IMPLEMENTATION,        "tfQMRgpu: Missing implementation at line %d!", 1},
    {TFQMRGPU_UNDOCUMENTED_ERROR,       "tfQMRgpu: Undocumented error at line %d!", 1},
    {TFQMRGPU_STATUS_LAUNCH_FAILED,     "tfQMRgpu: Device launch failed at line %d!", 1},
    {TFQMRGPU_DATALAYOUT_UNKNOWN,       "tfQMRgpu: Unknown data layout '0x%2.2x'!", 1},
    {TFQMRGPU_B_IS_NOT_SUBSET_OF_X,     "tfQMRgpu: B is not a subset of X in row %d!", 1},
    {TFQMRGPU_B_HAS_A_ZERO_COLUMN,      "tfQMRgpu: B has %d zero columns, will break!", 1},
    {TFQMRGPU_BLOCKSIZE_MISSING,        "tfQMRgpu: Missing blocksize %d x %d!", 2},
    {TFQMRGPU_TANSPOSITION_UNKNOWN,     "tfQMRgpu: Unknown transposition '%c' at line %d!", 3},
    {TFQMRGPU_VARIABLENAME_UNKNOWN,     "tfQMRgpu: Unknown variable name '%c' at line %d!", 3},
    {TFQMRGPU_PRECISION_MISSMATCH,      "tfQMRgpu: Missmatch in precision '%c' at line %d!", 3},
};
} // namespace

extern "C" char const* tfqmrgpuGetErrorString(tfqmrgpuStatus_t const status) {
    static char text[128];
    int const key  = status / TFQMRGPU_CODE_CHAR;
    int const rest = status - key * TFQMRGPU_CODE_CHAR;
    int const line = rest / TFQMRGPU_CODE_LINE;
    int const code = rest - line * TFQMRGPU_CODE_LINE;
    if (TFQMRGPU_STATUS_SUCCESS == code) { std::memset(text, 0, sizeof text); return text; }
    for (auto const& e : kTable) {
        if (e.code != code) continue;
        switch (e.args) {
            case 0: std::snprintf(text, sizeof text, "%s", e.text); break;
            case 1: std::snprintf(text, sizeof text, e.text, line); break;
            case 2: std::snprintf(text, sizeof text, e.text, key, line); break;
            default: std::snprintf(text, sizeof text, e.text, char(key), line); break;
        }
        return text;
    }
    std::snprintf(text, sizeof text, "tfQMRgpu: Unknown status= %d at line %d, key '%c', stat= %d!",
                  status, line, (key > 31) ? char(key) : '?', code);
    return text;
}

extern "C" tfqmrgpuStatus_t tfqmrgpuPrintError(tfqmrgpuStatus_t const status) {
    std::fflush(stdout);
    if (TFQMRGPU_STATUS_SUCCESS != status) std::printf("\n%s\n\n", tfqmrgpuGetErrorString(status));
    std::fflush(stdout);
    return TFQMRGPU_STATUS_SUCCESS;
}
