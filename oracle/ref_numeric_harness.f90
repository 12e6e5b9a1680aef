! ref_numeric_harness.f90 -- OUR OWN driver (test infrastructure).
! Links against the reference's src/numericUtilities.f95 compiled where it lies
! (oracle/Makefile, target `ref`) and prints bit patterns of its outputs so that
! tests/golden/ref_numeric.json can pin the oracle's restatement bit-for-bit.
! Output format: one record per line, "<tag> <n> <int32 bit patterns ...>".
program ref_numeric_harness
  use numericUtilities
  implicit none
  integer, parameter :: lobN(7) = (/ 2, 3, 5, 12, 64, 180, 299 /)
  integer :: i, j, n, k, g
  real, allocatable :: mus(:), w(:), P(:,:)
  real :: tmus(7), tableR(9), v
  real(8) :: tableD(9), cdf(8)
  integer :: state
  integer, parameter :: legN(3) = (/ 1, 12, 64 /)

  do i = 1, size(lobN)
    n = lobN(i)
    allocate(mus(n), w(n))
    call computeLobattoTerms(mus, w)
    write(*, '(A,1X,I0)', advance='no') 'lobatto_mus', n
    do j = 1, n; write(*, '(1X,I0)', advance='no') transfer(mus(j), 1); end do
    write(*, *)
    write(*, '(A,1X,I0)', advance='no') 'lobatto_w', n
    do j = 1, n; write(*, '(1X,I0)', advance='no') transfer(w(j), 1); end do
    write(*, *)
    deallocate(mus, w)
  end do

  tmus = (/ -1.0, -0.73, -0.1, 0.0, 0.31, 0.85, 1.0 /)
  do k = 1, 3
    n = legN(k)
    allocate(P(0:n, size(tmus)))
    P = computeLegendrePolynomials(n, tmus)
    write(*, '(A,1X,I0)', advance='no') 'legendre', n
    do j = 1, size(tmus)
      do i = 0, n; write(*, '(1X,I0)', advance='no') transfer(P(i, j), 1); end do
    end do
    write(*, *)
    deallocate(P)
  end do

  ! findIndex: an increasing table, probe values from a small LCG, every first guess
  tableR = (/ 0.0, 0.1, 0.25, 0.26, 0.5, 0.51, 0.75, 0.99, 1.0 /)
  tableD = dble(tableR)
  state = 12345
  do k = 1, 200
    state = mod(state * 1103 + 12347, 65536)
    v = real(state) / 65535.0
    if (k == 1) v = 0.0
    if (k == 2) v = 1.0
    if (k == 3) v = 0.25
    g = mod(k, 10)      ! 0 = no first guess
    if (g == 0) then
      write(*, '(A,1X,I0,1X,I0,1X,I0,1X,I0,1X,I0)') 'findindex', 0, transfer(v, 1), &
        findIndex(v, tableR), findIndex(dble(v), tableD), findIndex(v, tableD)
    else
      write(*, '(A,1X,I0,1X,I0,1X,I0,1X,I0,1X,I0)') 'findindex', g, transfer(v, 1), &
        findIndex(v, tableR, g), findIndex(dble(v), tableD, g), findIndex(v, tableD, g)
    end if
  end do

  cdf = (/ 0.05d0, 0.05d0, 0.2d0, 0.45d0, 0.450001d0, 0.8d0, 0.95d0, 1.0d0 /)
  state = 777
  do k = 1, 100
    state = mod(state * 1103 + 12347, 65536)
    v = real(state) / 65535.0
    if (k == 1) v = 0.0
    if (k == 2) v = 1.0
    if (k == 3) v = 0.05
    write(*, '(A,1X,I0,1X,I0)') 'findcdf', transfer(v, 1), findCDFIndex(v, cdf)
  end do
end program ref_numeric_harness
