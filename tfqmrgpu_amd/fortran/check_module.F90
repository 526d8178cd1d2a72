!! Test program of the Fortran boundary (own code; the check is the one of the reference's Fortran example,
!! example/tfqmrgpu_Fortran_example.F90:108-126: a dense non-symmetric system, A*X == B verified with matmul).
!! Blocks are Fortran arrays mat(s, r, inzb): fast index = column inside the block.  Exit code 0 on success.
program check_module
  use tfqmrgpu
  implicit none
  integer(kind=4), parameter :: mb = 4, ld = 16, ncol = 2
  integer(kind=4) :: rowPtrA(mb+1), colIndA(mb*mb), rowPtrX(mb+1), colIndX(mb*ncol)
  complex(kind=8) :: Amat(ld,ld,mb*mb), Bmat(ld,ld,mb*ncol), Xmat(ld,ld,mb*ncol)
  complex(kind=8) :: Afull(mb*ld, mb*ld), Bfull(mb*ld, ncol*ld), Xfull(mb*ld, ncol*ld)
  real(kind=8) :: re(ld,ld), im(ld,ld), residual, dev
  integer(kind=4) :: ib, jb, inz, i, iterations, ierr
  call random_seed()
  inz = 0
  do ib = 1, mb
    rowPtrA(ib) = inz + 1
    do jb = 1, mb
      inz = inz + 1; colIndA(inz) = jb
      call random_number(re); call random_number(im)
      Amat(:,:,inz) = cmplx(re - 0.5d0, im - 0.5d0, kind=8)
      if (ib == jb) then
        do i = 1, ld
          Amat(i,i,inz) = Amat(i,i,inz) + 12.d0
        enddo
      endif
      !! mat(s, r, inz) holds element (row r, column s) of the block
      Afull((ib-1)*ld+1:ib*ld, (jb-1)*ld+1:jb*ld) = transpose(Amat(:,:,inz))
    enddo
  enddo
  rowPtrA(mb+1) = inz + 1
  inz = 0
  do ib = 1, mb
    rowPtrX(ib) = inz + 1
    do jb = 1, ncol
      inz = inz + 1; colIndX(inz) = jb
      call random_number(re); call random_number(im)
      Bmat(:,:,inz) = cmplx(re, im, kind=8)
      Bfull((ib-1)*ld+1:ib*ld, (jb-1)*ld+1:jb*ld) = transpose(Bmat(:,:,inz))
    enddo
  enddo
  rowPtrX(mb+1) = inz + 1
  iterations = 300; residual = 1.d-11; ierr = 0
  call solve(mb, ld, rowPtrA, colIndA, Amat, 'n', rowPtrX, colIndX, Xmat, 'n', rowPtrX, colIndX, Bmat, 'n', &
             iterations, residual, 6, ierr)
  if (ierr /= 0) stop 2
  inz = 0
  do ib = 1, mb
    do jb = 1, ncol
      inz = inz + 1
      Xfull((ib-1)*ld+1:ib*ld, (jb-1)*ld+1:jb*ld) = transpose(Xmat(:,:,inz))
    enddo
  enddo
  dev = maxval(abs(matmul(Afull, Xfull) - Bfull))
  write(*, '(a,i0,a,es10.3,a,es10.3)') '# check_module: ', iterations, ' iterations, residual ', residual, ', max|A*X-B| = ', dev
  if (dev > 1.d-8) stop 3
  write(*, '(a)') '# check_module: OK'
end program
