!! The three cases of the reference's Fortran example (example/tfqmrgpu_Fortran_example.F90:22-46) run through this library's
!! module with the flags the example uses: A 'n', X 'n', B 't' -- and B, A, X all the SAME array, as there (A is its own
!! transposed right-hand side and the solution overwrites it).  Own code; the patterns are the example's data.
!!   1) one 32 x 32 block                      2) a full 4 x 4 pattern of 16 x 16 blocks
!!   3) a banded 4 x 4 pattern of 4 x 4 blocks: the solver works on the pattern-truncated product, so the dense check does not
!!      hold and the example only asks that it runs through (its lines 35-43)
!! Check of 1 and 2 (the example's lines 108-126): with full(s, r) = element (row r, column s), i.e. every dense matrix stored
!! transposed, matmul(Xfull, Afull) is (A X)^T and must equal Bfull built from the 't' blocks: max |A X - B| < 1e-8.
!! (Against the reference's CPU build this check FAILS with 1.3 | 1.3 | 1.1: its CPU multiply reads A untransposed, SURVEY App. B-1.)
!! Exit code 0 on success, 10 + case on a failure.
program check_example
  use tfqmrgpu, only: tfqmrgpu_bsrsv_complete
  implicit none
  real(kind=8) :: dev(3)
  integer :: fail
  fail = 0
  dev(1) = run_case(32, [1, 2], [1], .true.)
  dev(2) = run_case(16, [1, 5, 9, 13, 17], [1,2,3,4, 1,2,3,4, 1,2,3,4, 1,2,3,4], .true.)
  dev(3) = run_case(4, [1, 3, 6, 9, 11], [1,2, 1,2,3, 2,3,4, 3,4], .false.)
  write(*, '(a,3es10.2)') '# check_example: max|A*X - B| = ', dev
  if (.not. (dev(1) < 1.d-8)) fail = 11
  if (.not. (dev(2) < 1.d-8) .and. fail == 0) fail = 12
  if (dev(3) < 0.d0 .and. fail == 0) fail = 13          !! (case 3 has to run through: a negative value marks a solver error)
  if (fail /= 0) then
    write(*, '(a,i0)') '# check_example: FAILED, code ', fail
    call exit(fail)
  endif
  write(*, '(a)') '# check_example: OK'

contains

  !! dense image of a block list: full((jcol-1)*bd + s, (irow-1)*bd + r) = element (r, s) of block (irow, jcol), mat(s, r, inz)
  !! holding that element for flag 'n' and mat(r, s, inz) for 't'
  subroutine to_full(full, rowPtr, colInd, mat, flag)
    complex(kind=8), intent(out) :: full(:,:)
    integer(kind=4), intent(in) :: rowPtr(:), colInd(:)
    complex(kind=8), intent(in) :: mat(:,:,:)
    character, intent(in) :: flag
    integer :: irow, inz, bd, r0, s0
    bd = size(mat, 1)
    full = 0
    do irow = 1, size(rowPtr) - 1
      do inz = rowPtr(irow), rowPtr(irow+1) - 1
        r0 = (irow - 1)*bd; s0 = (colInd(inz) - 1)*bd
        if (flag == 'n') then
          full(s0+1:s0+bd, r0+1:r0+bd) = mat(:,:,inz)
        else
          full(s0+1:s0+bd, r0+1:r0+bd) = transpose(mat(:,:,inz))
        endif
      enddo
    enddo
  end subroutine

  real(kind=8) function run_case(bd, rowPtr, colInd, dense_check) result(dev)
    integer(kind=4), intent(in) :: bd, rowPtr(:), colInd(:)
    logical, intent(in) :: dense_check
    complex(kind=8), allocatable :: mat(:,:,:), Afull(:,:), Bfull(:,:), Xfull(:,:)
    real(kind=8), allocatable :: re(:,:,:), im(:,:,:)
    integer(kind=4) :: mb, nnzb, iterations, ierr
    real(kind=8) :: residual
    mb = size(rowPtr) - 1; nnzb = size(colInd)
    allocate(mat(bd,bd,nnzb), re(bd,bd,nnzb), im(bd,bd,nnzb))
    allocate(Afull(mb*bd, mb*bd), Bfull(mb*bd, mb*bd), Xfull(mb*bd, mb*bd))
    call random_number(re); call random_number(im)
    mat = cmplx(re, im, kind=8)
    call to_full(Afull, rowPtr, colInd, mat, 'n')
    call to_full(Bfull, rowPtr, colInd, mat, 't')
    iterations = 999; residual = 1.d-9; ierr = 1
    call tfqmrgpu_bsrsv_complete(mb, bd, rowPtr, colInd, mat, 'n', rowPtr, colInd, mat, 'n', rowPtr, colInd, mat, 't', &
                                 iterations, residual, 6, ierr)
    if (ierr /= 0) then
      write(*, '(a,i0,a,i0)') '# check_example: block size ', bd, ': solver status ', ierr
      dev = -1.d0
      if (dense_check) dev = huge(1.d0)
      return
    endif
    call to_full(Xfull, rowPtr, colInd, mat, 'n')
    dev = maxval(abs(matmul(Xfull, Afull) - Bfull))
    write(*, '(a,i0,a,i0,a,i0,a,es10.3,a,es10.3)') '# check_example: block size ', bd, ', ', nnzb, ' blocks: ', iterations, &
          ' iterations, residual ', residual, ', max|A*X - B| = ', dev
  end function
end program
