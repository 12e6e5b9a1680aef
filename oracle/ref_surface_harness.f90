! ref_surface_harness.f90 -- OUR OWN driver (test infrastructure).
! Links against the reference's src/ErrorMessages.f95, src/numericUtilities.f95 and src/surfaceProperties.f95,
! compiled where they lie (oracle/Makefile, target `ref`; none of the three needs a netcdf module), and prints
! what computeSurfaceReflectance returns for a list of positions, as bit patterns, so that
! tests/golden/ref_surface.json can pin the oracle's surface_reflectance / make_periodic bit for bit.
! The reflectance of patch (i, j) encodes its indices, so the value pins findIndex(makePeriodic(...)) too.
! Output: "surface <k> <numX> <numY>" then numX + numY lines of edge bit patterns (int64), then
!         "point <k> <x bits int64> <y bits int64> <reflectance bits int32>" per position.
program ref_surface_harness
  use ErrorMessages
  use surfaceProperties
  implicit none
  integer, parameter :: nPts = 600
  type(surfaceDescription) :: sfc
  type(ErrorMessage)       :: status
  real(8), allocatable :: xs(:), ys(:)
  real,    allocatable :: params(:, :, :)
  real(8) :: x, y, span
  real    :: r
  integer :: k, i, j, numX, numY
  integer(8) :: seed

  seed = 20241004_8
  do k = 1, 2
    if (k == 1) then   ! lower edges at 0, upper edges not representable in single precision
      numX = 7; numY = 5
      allocate(xs(numX), ys(numY))
      xs = (/ 0.0d0, 0.13d0, 0.27d0, 0.6d0, 0.61d0, 1.3d0, 1.7d0 /)
      ys = (/ 0.0d0, 0.4d0, 0.45d0, 0.9d0, 1.1d0 /)
    else               ! lower edges away from 0, uneven patches
      numX = 4; numY = 6
      allocate(xs(numX), ys(numY))
      xs = (/ 0.25d0, 0.7d0, 1.45d0, 2.05d0 /)
      ys = (/ -0.3d0, -0.1d0, 0.05d0, 0.35d0, 0.36d0, 0.77d0 /)
    end if
    allocate(params(1, numX - 1, numY - 1))
    do j = 1, numY - 1
      do i = 1, numX - 1
        params(1, i, j) = real(i + 10 * j) / 100.
      end do
    end do
    call initializeState(status)
    sfc = new_SurfaceDescription(params, xs, ys, status)
    if (stateIsFailure(status)) stop 1
    write(*, '(A,1X,I0,1X,I0,1X,I0)') 'surface', k, numX, numY
    do i = 1, numX; write(*, '(A,1X,I0)') 'xedge', transfer(xs(i), 1_8); end do
    do j = 1, numY; write(*, '(A,1X,I0)') 'yedge', transfer(ys(j), 1_8); end do
    do i = 1, nPts
      ! positions up to two periods outside the surface on either side (periodic images), plus a few special ones
      span = xs(numX) - xs(1)
      x = xs(1) + (lcg(seed) * 5.d0 - 2.d0) * span
      span = ys(numY) - ys(1)
      y = ys(1) + (lcg(seed) * 5.d0 - 2.d0) * span
      if (i == 1) x = xs(1)                            ! exactly on the lower edge (:224-225)
      if (i == 2) y = ys(1)
      if (i == 3) x = xs(1) - (xs(numX) - xs(1))       ! one period below the lower edge
      if (i == 4) x = xs(3)                            ! exactly on an interior edge
      if (i == 5) y = ys(2)
      if (i == 6) x = xs(numX) + 1.d-9                 ! just past the upper edge
      if (i == 7) y = ys(numY) - 1.d-12                ! just inside it
      r = computeSurfaceReflectance(sfc, x, y, 0.5, 0.5, 0., 0.)
      write(*, '(A,1X,I0,1X,I0,1X,I0,1X,I0)') 'point', k, transfer(x, 1_8), transfer(y, 1_8), transfer(r, 1)
    end do
    call finalize_SurfaceDescription(sfc)
    deallocate(xs, ys, params)
  end do
contains
  function lcg(s) result(u)   ! 48-bit linear congruential generator (drand48 constants), uniform in [0, 1)
    integer(8), intent(inout) :: s
    real(8) :: u
    s = iand(s * 25214903917_8 + 11_8, 281474976710655_8)
    u = real(s, 8) / 281474976710656.d0
  end function lcg
end program ref_surface_harness
