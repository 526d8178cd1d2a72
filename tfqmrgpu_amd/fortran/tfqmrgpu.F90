!! Fortran 90 module `tfqmrgpu` for the MI355X build of libtfQMRgpu.
!!
!! Same public names and argument lists as the reference module
!! (real-space/tfQMRgpu tfQMRgpu/include/tfqmrgpu_Fortran_module.F90:12-59 and its procedures):
!!   generic interfaces  create, destroy, free, set, get, solve, print_error
!!   quick start         tfqmrgpu_bsrsv_complete (square blocks), tfqmrgpu_bsrsv_rectangular (ldB >= ldA)
!!   constants           TFQMRGPU_HANDLE_KIND, TFQMRGPU_PLAN_KIND, cuda_stream_kind, TFQMRGPU_LAYOUT_*
!! so that `use tfqmrgpu` in existing codes keeps compiling.  Every procedure forwards to the F77-style
!! entry points name_ of tfqmrgpu_amd/csrc/tfq_fortran.c (status in the trailing ierr argument, Fortran
!! 1-based indices, device pointers as 8-byte integers).
module tfqmrgpu
  implicit none
  private

  integer, parameter, public :: TFQMRGPU_HANDLE_KIND = 8, TFQMRGPU_PLAN_KIND = 8, TFQMRGPU_PTR_KIND = 8
  integer, parameter, public :: cuda_stream_kind = 8          !! a hipStream_t travels in an 8-byte integer
  integer(kind=4), parameter, public :: TFQMRGPU_STATUS_SUCCESS  = 0
  integer(kind=4), parameter, public :: TFQMRGPU_LAYOUT_RRRRIIII = 15, TFQMRGPU_LAYOUT_RRIIRRII = 51, &
                                        TFQMRGPU_LAYOUT_RIRIRIRI = 85, TFQMRGPU_LAYOUT_DEFAULT  = 85

  public :: print_error, create, destroy, free, set, get, solve
  public :: tfqmrgpu_bsrsv_complete, tfqmrgpu_bsrsv_rectangular

  interface create
    module procedure new_handle, new_plan, new_workspace
  end interface
  interface destroy
    module procedure del_handle, del_plan
  end interface
  interface free
    module procedure del_workspace
  end interface
  interface set
    module procedure put_stream, put_buffer, put_matrix_c, put_matrix_z
  end interface
  interface get
    module procedure ask_stream, ask_buffer_size, ask_buffer, ask_matrix_c, ask_matrix_z, ask_info
  end interface
  interface solve
    module procedure run_solve, tfqmrgpu_bsrsv_complete, tfqmrgpu_bsrsv_rectangular
  end interface

contains

  subroutine print_error(status, ierr)
    integer(kind=4), intent(in)  :: status
    integer(kind=4), intent(out) :: ierr
    external :: tfqmrgpuprinterror
    call tfqmrgpuprinterror(status, ierr)
  end subroutine

  ! ---- handle, stream, workspace ------------------------------------------------------------------
  subroutine new_handle(handle, ierr)
    integer(kind=TFQMRGPU_HANDLE_KIND), intent(out) :: handle
    integer(kind=4), intent(out) :: ierr
    external :: tfqmrgpucreatehandle
    call tfqmrgpucreatehandle(handle, ierr)
  end subroutine

  subroutine del_handle(handle, ierr)
    integer(kind=TFQMRGPU_HANDLE_KIND), intent(inout) :: handle
    integer(kind=4), intent(out) :: ierr
    external :: tfqmrgpudestroyhandle
    call tfqmrgpudestroyhandle(handle, ierr)
  end subroutine

  subroutine put_stream(handle, streamId, ierr)
    integer(kind=TFQMRGPU_HANDLE_KIND), intent(in) :: handle
    integer(kind=cuda_stream_kind), intent(in) :: streamId
    integer(kind=4), intent(out) :: ierr
    external :: tfqmrgpusetstream
    call tfqmrgpusetstream(handle, streamId, ierr)
  end subroutine

  subroutine ask_stream(handle, streamId, ierr)
    integer(kind=TFQMRGPU_HANDLE_KIND), intent(in) :: handle
    integer(kind=cuda_stream_kind), intent(out) :: streamId
    integer(kind=4), intent(out) :: ierr
    external :: tfqmrgpugetstream
    call tfqmrgpugetstream(handle, streamId, ierr)
  end subroutine

  subroutine new_workspace(pBuffer, pBufferSizeInBytes, ierr)
    integer(kind=TFQMRGPU_PTR_KIND), intent(inout) :: pBuffer
    integer(kind=8), intent(in) :: pBufferSizeInBytes
    integer(kind=4), intent(out) :: ierr
    external :: tfqmrgpucreateworkspace
    call tfqmrgpucreateworkspace(pBuffer, pBufferSizeInBytes, ierr)
  end subroutine

  subroutine del_workspace(pBuffer, ierr)
    integer(kind=TFQMRGPU_PTR_KIND), intent(inout) :: pBuffer
    integer(kind=4), intent(out) :: ierr
    external :: tfqmrgpudestroyworkspace
    call tfqmrgpudestroyworkspace(pBuffer, ierr)
  end subroutine

  ! ---- bsrsv: plan, buffer, matrices, solve, info ---------------------------------------------------
  subroutine new_plan(handle, plan, mb, bsrRowPtrA, nnzbA, bsrColIndA, bsrRowPtrX, nnzbX, bsrColIndX, &
                      bsrRowPtrB, nnzbB, bsrColIndB, echo, ierr)
    integer(kind=TFQMRGPU_HANDLE_KIND), intent(in) :: handle
    integer(kind=TFQMRGPU_PLAN_KIND), intent(out) :: plan
    integer(kind=4), intent(in) :: mb, nnzbA, nnzbX, nnzbB, echo
    integer(kind=4), intent(in) :: bsrRowPtrA(*), bsrColIndA(*), bsrRowPtrX(*), bsrColIndX(*), bsrRowPtrB(*), bsrColIndB(*)
    integer(kind=4), intent(out) :: ierr
    external :: tfqmrgpu_bsrsv_createplan
    call tfqmrgpu_bsrsv_createplan(handle, plan, mb, bsrRowPtrA, nnzbA, bsrColIndA, bsrRowPtrX, nnzbX, bsrColIndX, &
                                   bsrRowPtrB, nnzbB, bsrColIndB, echo, ierr)
  end subroutine

  subroutine del_plan(handle, plan, ierr)
    integer(kind=TFQMRGPU_HANDLE_KIND), intent(in) :: handle
    integer(kind=TFQMRGPU_PLAN_KIND), intent(inout) :: plan
    integer(kind=4), intent(out) :: ierr
    external :: tfqmrgpu_bsrsv_destroyplan
    call tfqmrgpu_bsrsv_destroyplan(handle, plan, ierr)
  end subroutine

  subroutine ask_buffer_size(handle, plan, ldA, blockDim, ldB, RhsBlockDim, prec, pBufferSizeInBytes, ierr)
    integer(kind=TFQMRGPU_HANDLE_KIND), intent(in) :: handle
    integer(kind=TFQMRGPU_PLAN_KIND), intent(inout) :: plan
    integer(kind=4), intent(in) :: ldA, blockDim, ldB, RhsBlockDim
    character, intent(in) :: prec
    integer(kind=8), intent(out) :: pBufferSizeInBytes
    integer(kind=4), intent(out) :: ierr
    external :: tfqmrgpu_bsrsv_buffersize
    call tfqmrgpu_bsrsv_buffersize(handle, plan, ldA, blockDim, ldB, RhsBlockDim, prec, pBufferSizeInBytes, ierr)
  end subroutine

  subroutine put_buffer(handle, plan, pBuffer, ierr)
    integer(kind=TFQMRGPU_HANDLE_KIND), intent(in) :: handle
    integer(kind=TFQMRGPU_PLAN_KIND), intent(inout) :: plan
    integer(kind=TFQMRGPU_PTR_KIND), intent(in) :: pBuffer
    integer(kind=4), intent(out) :: ierr
    external :: tfqmrgpu_bsrsv_setbuffer
    call tfqmrgpu_bsrsv_setbuffer(handle, plan, pBuffer, ierr)
  end subroutine

  subroutine ask_buffer(handle, plan, pBuffer, ierr)
    integer(kind=TFQMRGPU_HANDLE_KIND), intent(in) :: handle
    integer(kind=TFQMRGPU_PLAN_KIND), intent(in) :: plan
    integer(kind=TFQMRGPU_PTR_KIND), intent(inout) :: pBuffer
    integer(kind=4), intent(out) :: ierr
    external :: tfqmrgpu_bsrsv_getbuffer
    call tfqmrgpu_bsrsv_getbuffer(handle, plan, pBuffer, ierr)
  end subroutine

  subroutine put_matrix_c(handle, plan, var, val, ld, d2, trans, layout, ierr)
    integer(kind=TFQMRGPU_HANDLE_KIND), intent(in) :: handle
    integer(kind=TFQMRGPU_PLAN_KIND), intent(inout) :: plan
    character, intent(in) :: var, trans
    complex(kind=4), intent(in) :: val(*)
    integer(kind=4), intent(in) :: ld, d2, layout
    integer(kind=4), intent(out) :: ierr
    external :: tfqmrgpu_bsrsv_setmatrix_c
    call tfqmrgpu_bsrsv_setmatrix_c(handle, plan, var, val, ld, d2, trans, layout, ierr)
  end subroutine

  subroutine put_matrix_z(handle, plan, var, val, ld, d2, trans, layout, ierr)
    integer(kind=TFQMRGPU_HANDLE_KIND), intent(in) :: handle
    integer(kind=TFQMRGPU_PLAN_KIND), intent(inout) :: plan
    character, intent(in) :: var, trans
    complex(kind=8), intent(in) :: val(*)
    integer(kind=4), intent(in) :: ld, d2, layout
    integer(kind=4), intent(out) :: ierr
    external :: tfqmrgpu_bsrsv_setmatrix_z
    call tfqmrgpu_bsrsv_setmatrix_z(handle, plan, var, val, ld, d2, trans, layout, ierr)
  end subroutine

  subroutine ask_matrix_c(handle, plan, var, val, ld, d2, trans, layout, ierr)
    integer(kind=TFQMRGPU_HANDLE_KIND), intent(in) :: handle
    integer(kind=TFQMRGPU_PLAN_KIND), intent(in) :: plan
    character, intent(in) :: var, trans
    complex(kind=4), intent(out) :: val(*)
    integer(kind=4), intent(in) :: ld, d2, layout
    integer(kind=4), intent(out) :: ierr
    external :: tfqmrgpu_bsrsv_getmatrix_c
    call tfqmrgpu_bsrsv_getmatrix_c(handle, plan, var, val, ld, d2, trans, layout, ierr)
  end subroutine

  subroutine ask_matrix_z(handle, plan, var, val, ld, d2, trans, layout, ierr)
    integer(kind=TFQMRGPU_HANDLE_KIND), intent(in) :: handle
    integer(kind=TFQMRGPU_PLAN_KIND), intent(in) :: plan
    character, intent(in) :: var, trans
    complex(kind=8), intent(out) :: val(*)
    integer(kind=4), intent(in) :: ld, d2, layout
    integer(kind=4), intent(out) :: ierr
    external :: tfqmrgpu_bsrsv_getmatrix_z
    call tfqmrgpu_bsrsv_getmatrix_z(handle, plan, var, val, ld, d2, trans, layout, ierr)
  end subroutine

  subroutine run_solve(handle, plan, threshold, maxIterations, ierr)
    integer(kind=TFQMRGPU_HANDLE_KIND), intent(in) :: handle
    integer(kind=TFQMRGPU_PLAN_KIND), intent(inout) :: plan
    real(kind=8), intent(in) :: threshold
    integer(kind=4), intent(in) :: maxIterations
    integer(kind=4), intent(out) :: ierr
    external :: tfqmrgpu_bsrsv_solve
    call tfqmrgpu_bsrsv_solve(handle, plan, threshold, maxIterations, ierr)
  end subroutine

  subroutine ask_info(handle, plan, residual_reached, iterations_needed, flops_performed, flops_performed_all, ierr)
    integer(kind=TFQMRGPU_HANDLE_KIND), intent(in) :: handle
    integer(kind=TFQMRGPU_PLAN_KIND), intent(inout) :: plan
    real(kind=8), intent(out) :: residual_reached, flops_performed, flops_performed_all
    integer(kind=4), intent(out) :: iterations_needed
    integer(kind=4), intent(out) :: ierr
    external :: tfqmrgpu_bsrsv_getinfo
    call tfqmrgpu_bsrsv_getinfo(handle, plan, residual_reached, iterations_needed, flops_performed, flops_performed_all, ierr)
  end subroutine

  ! ---- quick start: all steps in one call (reference module :294-421) -----------------------------------
  !! blocks are Fortran arrays Amat(ldA,ldA,nnzbA), Xmat(ldB,ldA,nnzbX), Bmat(ldB,ldA,nnzbB): the fast index is the
  !! column inside a block, i.e. the same bytes as C row-major blocks [ldA][ldB]; RowPtr/ColInd are 1-based.
  !! o: Fortran unit for messages (0: mute).  On exit iterations/residual hold what was needed/reached.
  subroutine tfqmrgpu_bsrsv_rectangular(mb, ldA, ldB, rowPtrA, colIndA, Amat, transA, rowPtrX, colIndX, Xmat, transX, &
                                        rowPtrB, colIndB, Bmat, transB, iterations, residual, o, ierr)
    integer(kind=4), intent(in) :: mb, ldA, ldB
    integer(kind=4), intent(in) :: rowPtrA(:), rowPtrX(:), rowPtrB(:), colIndA(:), colIndX(:), colIndB(:)
    character, intent(in) :: transA, transX, transB
    complex(kind=8), intent(in)  :: Amat(ldA,ldA,*), Bmat(ldB,ldA,*)
    complex(kind=8), intent(out) :: Xmat(ldB,ldA,*)
    integer(kind=4), intent(inout) :: iterations
    real(kind=8), intent(inout) :: residual
    integer(kind=4), intent(in) :: o
    integer(kind=4), intent(inout) :: ierr

    integer(kind=TFQMRGPU_HANDLE_KIND) :: handle
    integer(kind=TFQMRGPU_PLAN_KIND) :: plan
    integer(kind=TFQMRGPU_PTR_KIND) :: buffer
    integer(kind=cuda_stream_kind) :: stream
    integer(kind=8) :: nbytes
    integer(kind=4) :: stat, dummy, needed, nA, nX, nB, echo
    real(kind=8) :: reached, flops, flops_all

    echo = 0; if (ierr /= 0) echo = 9    !! a nonzero ierr on entry asks for a verbose run
    nA = size(colIndA); nX = size(colIndX); nB = size(colIndB)
    handle = 0; plan = 0; buffer = 0; stream = 0
    call create(handle, stat);                       if (failed('create handle')) return
    call set(handle, stream, stat);                  if (failed('set stream')) return
    call create(handle, plan, mb, rowPtrA, nA, colIndA, rowPtrX, nX, colIndX, rowPtrB, nB, colIndB, echo, stat)
                                                     if (failed('create plan')) return
    call get(handle, plan, ldA, ldA, ldB, ldB, 'z', nbytes, stat); if (failed('buffer size')) return
    call create(buffer, nbytes, stat);               if (failed('create workspace')) return
    call set(handle, plan, buffer, stat);            if (failed('set buffer')) return
    !! specific procedures + sequence association: the block arrays are contiguous
    call put_matrix_z(handle, plan, 'A', Amat(1,1,1), ldA, ldA, transA, TFQMRGPU_LAYOUT_RIRIRIRI, stat); if (failed('set A')) return
    call put_matrix_z(handle, plan, 'B', Bmat(1,1,1), ldB, ldA, transB, TFQMRGPU_LAYOUT_RIRIRIRI, stat); if (failed('set B')) return
    call solve(handle, plan, residual, iterations, stat)
    ierr = stat                                      !! 0 converged, 9 out of iterations, 6 breakdown
    call get(handle, plan, reached, needed, flops, flops_all, stat); if (failed('get info')) return
    if (o > 0) write(o, '(a,i0,a,es10.3,a,f0.3,a)') '# tfQMRgpu needed ', needed, ' iterations to converge to ', &
                                                    reached, ' using ', flops*1e-9, ' GFlop'
    residual = reached; iterations = needed
    stat = 0
    call ask_matrix_z(handle, plan, 'X', Xmat(1,1,1), ldB, ldA, transX, TFQMRGPU_LAYOUT_RIRIRIRI, stat); if (failed('get X')) return
    call free(buffer, dummy)
    call destroy(handle, plan, dummy)
    call destroy(handle, dummy)
  contains
    logical function failed(what)
      character(len=*), intent(in) :: what
      failed = (stat /= 0)
      if (.not. failed) return
      ierr = stat
      if (o > 0) write(o, '(3a,i0)') '# tfqmrgpu_bsrsv: ', what, ' returned status ', stat
      call print_error(stat, dummy)
      if (buffer /= 0) call free(buffer, dummy)
      if (plan /= 0) call destroy(handle, plan, dummy)
      if (handle /= 0) call destroy(handle, dummy)
    end function
  end subroutine

  subroutine tfqmrgpu_bsrsv_complete(mb, ldA, rowPtrA, colIndA, Amat, transA, rowPtrX, colIndX, Xmat, transX, &
                                     rowPtrB, colIndB, Bmat, transB, iterations, residual, o, ierr)
    integer(kind=4), intent(in) :: mb, ldA
    integer(kind=4), intent(in) :: rowPtrA(:), rowPtrX(:), rowPtrB(:), colIndA(:), colIndX(:), colIndB(:)
    character, intent(in) :: transA, transX, transB
    complex(kind=8), intent(in)  :: Amat(ldA,ldA,*), Bmat(ldA,ldA,*)
    complex(kind=8), intent(out) :: Xmat(ldA,ldA,*)
    integer(kind=4), intent(inout) :: iterations
    real(kind=8), intent(inout) :: residual
    integer(kind=4), intent(in) :: o
    integer(kind=4), intent(inout) :: ierr
    call tfqmrgpu_bsrsv_rectangular(mb, ldA, ldA, rowPtrA, colIndA, Amat, transA, rowPtrX, colIndX, Xmat, transX, &
                                    rowPtrB, colIndB, Bmat, transB, iterations, residual, o, ierr)
  end subroutine

end module tfqmrgpu
