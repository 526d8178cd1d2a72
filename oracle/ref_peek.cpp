// TEST INFRASTRUCTURE (oracle/_ref only): lets the golden-vector generator read the analysis results
// out of the REFERENCE's opaque plan (bsrsv_plan_t, tfqmrgpu_plan.hxx:9-55).  Compiled against the
// reference headers where they lie; exists only in this container.
#include "tfqmrgpu.hxx"
#include "tfqmrgpu_plan.hxx"

extern "C" {
void refpeek_sizes(void* plan, uint64_t* nPairs, uint32_t* nCols, uint32_t* nnzbX, uint32_t* nnzbB) {
    auto p = (bsrsv_plan_t*)plan;
    *nPairs = p->pairs.size() / 2; *nCols = p->nCols; *nnzbX = uint32_t(p->colindx.size()); *nnzbB = uint32_t(p->subset.size());
}
void refpeek_copy(void* plan, uint32_t* pairs, uint32_t* starts, uint32_t* subset, uint16_t* colindx, int32_t* orig) {
    auto p = (bsrsv_plan_t*)plan;
    std::copy(p->pairs.begin(), p->pairs.end(), pairs);
    std::copy(p->starts.begin(), p->starts.end(), starts);
    std::copy(p->subset.begin(), p->subset.end(), subset);
    std::copy(p->colindx.begin(), p->colindx.end(), colindx);
    std::copy(p->original_bsrColIndX.begin(), p->original_bsrColIndX.end(), orig);
}
}
