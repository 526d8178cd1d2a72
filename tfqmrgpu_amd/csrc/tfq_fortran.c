/* Fortran-callable entry points of libtfQMRgpu (F77 calling convention: lower-case name with a
 * trailing underscore, every argument by reference, the status as the last argument).
 *
 * Replaces the 18 symbols of real-space/tfQMRgpu tfQMRgpu/source/tfqmrgpu_Fortran_wrappers.c:58-187.
 * Behaviour kept: createplan uses Fortran (1-based) indices and clears the plan before the call;
 * createworkspace always allocates device memory; the _c_/_z_ variants fix the precision character;
 * handles, plans and device pointers travel as 8-byte integers on the Fortran side.
 */
#include <stddef.h>
#include <stdint.h>

#include "tfqmrgpu.h"

#define FWRAP(name) void name##_

FWRAP(tfqmrgpuprinterror)(tfqmrgpuStatus_t const *status, tfqmrgpuStatus_t *stat) {
    *stat = tfqmrgpuPrintError(*status);
}

FWRAP(tfqmrgpucreatehandle)(tfqmrgpuHandle_t *handle, tfqmrgpuStatus_t *stat) {
    *handle = NULL;
    *stat = tfqmrgpuCreateHandle(handle);
}

FWRAP(tfqmrgpudestroyhandle)(tfqmrgpuHandle_t *handle, tfqmrgpuStatus_t *stat) {
    *stat = tfqmrgpuDestroyHandle(*handle);
    *handle = NULL;
}

FWRAP(tfqmrgpusetstream)(tfqmrgpuHandle_t const *handle, int64_t const *streamId, tfqmrgpuStatus_t *stat) {
    *stat = tfqmrgpuSetStream(*handle, (tfqmrgpuStream_t)(intptr_t)(*streamId));
}

FWRAP(tfqmrgpugetstream)(tfqmrgpuHandle_t const *handle, int64_t *streamId, tfqmrgpuStatus_t *stat) {
    tfqmrgpuStream_t s = 0;
    *stat = tfqmrgpuGetStream(*handle, &s);
    *streamId = (int64_t)(intptr_t)s;
}

FWRAP(tfqmrgpu_bsrsv_createplan)(tfqmrgpuHandle_t const *handle, tfqmrgpuBsrsvPlan_t *plan, int32_t const *mb,
    int32_t const *bsrRowPtrA, int32_t const *nnzbA, int32_t const *bsrColIndA,
    int32_t const *bsrRowPtrX, int32_t const *nnzbX, int32_t const *bsrColIndX,
    int32_t const *bsrRowPtrB, int32_t const *nnzbB, int32_t const *bsrColIndB,
    int32_t const *echo, tfqmrgpuStatus_t *stat)
{
    *plan = NULL;
    *stat = tfqmrgpu_bsrsv_createPlan(*handle, plan, *mb,
                bsrRowPtrA, *nnzbA, bsrColIndA, bsrRowPtrX, *nnzbX, bsrColIndX, bsrRowPtrB, *nnzbB, bsrColIndB,
                1 /* Fortran index offset */, *echo);
    if (TFQMRGPU_STATUS_SUCCESS != *stat) tfqmrgpuPrintError(*stat);
}

FWRAP(tfqmrgpu_bsrsv_destroyplan)(tfqmrgpuHandle_t const *handle, tfqmrgpuBsrsvPlan_t *plan, tfqmrgpuStatus_t *stat) {
    *stat = tfqmrgpu_bsrsv_destroyPlan(*handle, *plan);
    *plan = NULL;
}

FWRAP(tfqmrgpu_bsrsv_buffersize)(tfqmrgpuHandle_t const *handle, tfqmrgpuBsrsvPlan_t const *plan,
    int32_t const *ldA, int32_t const *blockDim, int32_t const *ldB, int32_t const *RhsBlockDim,
    char const *precision, size_t *pBufferSizeInBytes, tfqmrgpuStatus_t *stat)
{
    *stat = tfqmrgpu_bsrsv_bufferSize(*handle, *plan, *ldA, *blockDim, *ldB, *RhsBlockDim, *precision, pBufferSizeInBytes);
}

FWRAP(tfqmrgpucreateworkspace)(void **pBuffer, size_t const *pBufferSizeInBytes, tfqmrgpuStatus_t *stat) {
    *stat = tfqmrgpuCreateWorkspace(pBuffer, *pBufferSizeInBytes, 'd');
}

FWRAP(tfqmrgpudestroyworkspace)(void **pBuffer, tfqmrgpuStatus_t *stat) {
    *stat = tfqmrgpuDestroyWorkspace(*pBuffer);
}

FWRAP(tfqmrgpu_bsrsv_setbuffer)(tfqmrgpuHandle_t const *handle, tfqmrgpuBsrsvPlan_t const *plan, void *const *pBuffer, tfqmrgpuStatus_t *stat) {
    *stat = tfqmrgpu_bsrsv_setBuffer(*handle, *plan, *pBuffer);
}

FWRAP(tfqmrgpu_bsrsv_getbuffer)(tfqmrgpuHandle_t const *handle, tfqmrgpuBsrsvPlan_t const *plan, void **pBuffer, tfqmrgpuStatus_t *stat) {
    *stat = tfqmrgpu_bsrsv_getBuffer(*handle, *plan, pBuffer);
}

FWRAP(tfqmrgpu_bsrsv_setmatrix_c)(tfqmrgpuHandle_t const *handle, tfqmrgpuBsrsvPlan_t const *plan, char const *var,
    float const *val, int32_t const *ld, int32_t const *d2, char const *trans, tfqmrgpuDataLayout_t const *layout, tfqmrgpuStatus_t *stat)
{
    *stat = tfqmrgpu_bsrsv_setMatrix(*handle, *plan, *var, (void const*)val, 'c', *ld, *d2, *trans, *layout);
}

FWRAP(tfqmrgpu_bsrsv_setmatrix_z)(tfqmrgpuHandle_t const *handle, tfqmrgpuBsrsvPlan_t const *plan, char const *var,
    double const *val, int32_t const *ld, int32_t const *d2, char const *trans, tfqmrgpuDataLayout_t const *layout, tfqmrgpuStatus_t *stat)
{
    *stat = tfqmrgpu_bsrsv_setMatrix(*handle, *plan, *var, (void const*)val, 'z', *ld, *d2, *trans, *layout);
}

FWRAP(tfqmrgpu_bsrsv_getmatrix_c)(tfqmrgpuHandle_t const *handle, tfqmrgpuBsrsvPlan_t const *plan, char const *var,
    float *val, int32_t const *ld, int32_t const *d2, char const *trans, tfqmrgpuDataLayout_t const *layout, tfqmrgpuStatus_t *stat)
{
    *stat = tfqmrgpu_bsrsv_getMatrix(*handle, *plan, *var, (void*)val, 'c', *ld, *d2, *trans, *layout);
}

FWRAP(tfqmrgpu_bsrsv_getmatrix_z)(tfqmrgpuHandle_t const *handle, tfqmrgpuBsrsvPlan_t const *plan, char const *var,
    double *val, int32_t const *ld, int32_t const *d2, char const *trans, tfqmrgpuDataLayout_t const *layout, tfqmrgpuStatus_t *stat)
{
    *stat = tfqmrgpu_bsrsv_getMatrix(*handle, *plan, *var, (void*)val, 'z', *ld, *d2, *trans, *layout);
}

FWRAP(tfqmrgpu_bsrsv_solve)(tfqmrgpuHandle_t const *handle, tfqmrgpuBsrsvPlan_t const *plan,
    double const *threshold, int32_t const *maxIterations, tfqmrgpuStatus_t *stat)
{
    *stat = tfqmrgpu_bsrsv_solve(*handle, *plan, *threshold, *maxIterations);
}

FWRAP(tfqmrgpu_bsrsv_getinfo)(tfqmrgpuHandle_t const *handle, tfqmrgpuBsrsvPlan_t const *plan, double *residuum_reached,
    int32_t *iterations_needed, double *flops_performed, double *flops_performed_all, tfqmrgpuStatus_t *stat)
{
    *stat = tfqmrgpu_bsrsv_getInfo(*handle, *plan, residuum_reached, iterations_needed, flops_performed, flops_performed_all);
}
