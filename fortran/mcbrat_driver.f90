! mcbrat_driver.f90 -- a small stand-in for Drivers/monteCarloDriver.f95 around the HIP integrator.
!
! Reads the reference's five namelists from the file named on the command line (same group and
! variable names and defaults, monteCarloDriver.f95:58-121), traces numBatches batches through
! mcbrat_hip_integrator, forms mean / standard error from the batch moments as :1188-1228 and
! writes the flux file in the layout of writeResults_ASCII (:1376-1399).
!
! Domains: the reference reads NetCDF (.dom / SSP tables); no NetCDF library exists in this
! image, so physDomainFile names either a built-in I3RC generator
!   builtin:i3rcStepCloud | builtin:i3rcStepCloudConservative | builtin:planeParallel
! (Domain-Files/i3rcStepCloud.f95, planeParallel.f95) or a flat binary written by
! mcbrat3d_amd.flatdomain.write_flat_domain (any domain the Python host layer can build).
program mcbrat_driver
  use mcbrat_hip_integrator
  implicit none
  ! --- namelist variables (names and defaults of the reference driver) ---
  real     :: solarMu = 1., solarAzimuth = 0., LW_flag = -1.
  real(8)  :: surfaceTemp = 300.0
  integer, parameter :: maxNumRad = 648
  real     :: intensityMus(maxNumRad) = 0., intensityPhis(maxNumRad) = 0.
  logical  :: angleFill = .false., calcRayl = .true.
  real, dimension(3) :: thetaFill = -1., phiFill = -1.
  integer  :: numLambda = 1
  integer(8) :: numPhotonsPerBatch = 0
  integer  :: numBatches = 100, iseed = 10, nPhaseIntervals = 10001
  logical  :: useRayTracing = .true., useRussianRoulette = .true.
  logical  :: useHybridPhaseFunsForIntenCalcs = .false.
  real     :: hybridPhaseFunWidth = 7.
  integer  :: numOrdersOrigPhaseFunIntenCalcs = 0
  logical  :: useRussianRouletteForIntensity = .true.
  real     :: zetaMin = 0.3
  logical  :: limitIntensityContributions = .false.
  real     :: maxIntensityContribution = 77.
  logical  :: reportVolumeAbsorption = .false., reportAbsorptionProfile = .false.
  logical  :: recScatOrd = .false.
  integer  :: numRecScatOrd = 0
  character(len=256) :: auxhist01_radFile = "", auxhist01_fluxFile = ""
  character(len=256) :: solarSourceFile = "", instrResponseFile = "", physDomainFile = ""
  character(len=256), dimension(4) :: SSPfilename = ""
  character(len=256) :: outputFluxFile = "", outputRadFile = "", outputAbsProfFile = "", &
                        outputAbsVolumeFile = "", outputNetcdfFile = ""
  namelist /radiativeTransfer/ solarMu, solarAzimuth, surfaceTemp, intensityMus, intensityPhis, &
                               angleFill, thetaFill, phiFill, LW_flag, numLambda, calcRayl
  namelist /monteCarlo/        numPhotonsPerBatch, numBatches, iseed, nPhaseIntervals
  namelist /algorithms/        useRayTracing, useRussianRoulette, useHybridPhaseFunsForIntenCalcs, &
                               hybridPhaseFunWidth, numOrdersOrigPhaseFunIntenCalcs, &
                               useRussianRouletteForIntensity, zetaMin, limitIntensityContributions, &
                               maxIntensityContribution
  namelist /output/            reportVolumeAbsorption, reportAbsorptionProfile, recScatOrd, numRecScatOrd, &
                               auxhist01_fluxFile, auxhist01_radFile
  namelist /fileNames/         solarSourceFile, instrResponseFile, SSPfilename, physDomainFile, &
                               outputRadFile, outputFluxFile, outputAbsProfFile, outputAbsVolumeFile, &
                               outputNetcdfFile
  ! --- locals ---
  character(len=256) :: namelistFileName
  type(integrator)   :: mcIntegrator
  integer :: nx, ny, nz, nc, ierr, i, j, k, c, nSteps, nEntries, ncol
  integer(8) :: M, nDone, nvox
  real(8) :: albedo, solarFlux, totalNumPhotons, batchesCompleted
  real(8), allocatable :: xPosition(:), yPosition(:), zPosition(:), totalExt(:,:,:), cumExt(:,:,:,:), ssa(:,:,:,:)
  integer, allocatable :: phaseFuncI(:,:,:,:)
  real,    allocatable :: table(:,:)
  real(8), allocatable :: moments(:)
  real(8), allocatable :: meanStats(:,:), fluxUpStats(:,:,:), fluxDownStats(:,:,:), fluxAbsorbedStats(:,:,:), &
                          absorbedProfileStats(:,:), RadianceStats(:,:,:,:)
  real,    allocatable :: forwardTable(:,:), legendreCoefficients(:)
  integer :: numRadDir, off
  logical :: computeIntensity
  real :: t0, t1

  if (command_argument_count() < 1) stop "usage: mcbrat_driver <namelist file>"
  call get_command_argument(1, namelistFileName)
  open (unit = 1, file = trim(namelistFileName), status = 'old')
  read (1, nml = radiativeTransfer); rewind(1)
  read (1, nml = monteCarlo); rewind(1)
  read (1, nml = algorithms); rewind(1)
  read (1, nml = output); rewind(1)
  read (1, nml = fileNames); close (1)
  if (numPhotonsPerBatch <= 0) stop "must specify numPhotonsPerBatch"
  if (len_trim(physDomainFile) == 0) stop "must specify physDomainFile"
  solarFlux = 1.0_8

  call cpu_time(t0)
  if (physDomainFile(1:8) == "builtin:") then
    call builtinDomain(trim(physDomainFile(9:)))
  else
    call readFlatDomain(trim(physDomainFile))
  end if

  mcIntegrator = new_Integrator(xPosition, yPosition, zPosition, 0, ierr); call check("new_Integrator")
  call setOpticalProperties(mcIntegrator, totalExt, cumExt, ssa, phaseFuncI, albedo, ierr); call check("setOpticalProperties")
  call specifyParameters(mcIntegrator, useRayTracing, useRussianRoulette, LW_flag, ierr); call check("specifyParameters")
  call loadTables()
  ! intensity directions: entries with |mu| > 0, only if a file will hold them (monteCarloDriver.f95:279-282, :547-594)
  numRadDir = count(abs(intensityMus(:)) > 0.)
  computeIntensity = numRadDir > 0 .and. len_trim(outputRadFile) > 0
  if (computeIntensity) then
    if (.not. allocated(legendreCoefficients)) stop "intensity needs the phase function of a builtin domain"
    allocate(forwardTable(max(nPhaseIntervals, 9001), 1))
    call forwardTableLegendre(legendreCoefficients, forwardTable(:, 1), ierr); call check("forwardTableLegendre")
    call specifyIntensity(mcIntegrator, pack(intensityMus, abs(intensityMus) > 0.), pack(intensityPhis, abs(intensityMus) > 0.), &
                          useRussianRouletteForIntensity, zetaMin, .false., numOrdersOrigPhaseFunIntenCalcs, &
                          limitIntensityContributions, maxIntensityContribution, ierr)
    call check("specifyParameters")
    call setForwardTable(mcIntegrator, 1, forwardTable, forwardTable, ierr); call check("setForwardTable")
  end if
  call setSolarSource(mcIntegrator, solarMu, solarAzimuth, ierr); call check("setSolarSource")
  call resetMoments(mcIntegrator, ierr); call check("resetMoments")
  call cpu_time(t1)
  print '(A,F8.3,A)', " setup ", t1 - t0, " s"

  ! worker loop (monteCarloDriver.f95:889-1085): all batches in one call; per-batch
  ! reportResults and moment accumulation (:1008-1050) happen on the device
  call computeRadiativeTransfer(mcIntegrator, int(iseed, 8), 0_8, numPhotonsPerBatch, numBatches, nDone, ierr)
  call check("computeRadiativeTransfer")
  print '(A,I14,A,F10.3,A,ES10.3,A)', " traced ", nDone, " photons, kernel ", lastTraceMilliseconds(mcIntegrator), &
        " ms = ", real(nDone) / (1.e-3 * lastTraceMilliseconds(mcIntegrator)), " photons/s"
  print '(A,I14)', " photons dropped by a loop bound (the reference's nBad): ", numBadPhotons(mcIntegrator)

  ! moments -> mean and standard error (:1188-1228)
  M = momentsLength(mcIntegrator)
  allocate(moments(8 + 2 * M))
  call getMoments(mcIntegrator, moments, ierr); call check("getMoments")
  totalNumPhotons = moments(1); batchesCompleted = moments(2)
  ncol = nx * ny
  allocate(meanStats(3, 2), fluxUpStats(nx, ny, 2), fluxDownStats(nx, ny, 2), fluxAbsorbedStats(nx, ny, 2), &
           absorbedProfileStats(nz, 2))
  do k = 1, 2
    meanStats(:, k) = moments(8 + (k-1)*M + 1 : 8 + (k-1)*M + 3)
    fluxUpStats(:, :, k)       = reshape(moments(8 + (k-1)*M + 3 + 1          : 8 + (k-1)*M + 3 + ncol),   (/ nx, ny /))
    fluxDownStats(:, :, k)     = reshape(moments(8 + (k-1)*M + 3 + ncol + 1   : 8 + (k-1)*M + 3 + 2*ncol), (/ nx, ny /))
    fluxAbsorbedStats(:, :, k) = reshape(moments(8 + (k-1)*M + 3 + 2*ncol + 1 : 8 + (k-1)*M + 3 + 3*ncol), (/ nx, ny /))
    absorbedProfileStats(:, k) = moments(8 + (k-1)*M + 3 + 3*ncol + 1 : 8 + (k-1)*M + 3 + 3*ncol + nz)
  end do
  call momentsToStats1(meanStats); call momentsToStats2(fluxUpStats); call momentsToStats2(fluxDownStats)
  call momentsToStats2(fluxAbsorbedStats); call momentsToStats1(absorbedProfileStats)

  print '(A,3(2X,F9.6,A,F9.6))', " mean flux up/down/absorbed:", meanStats(1,1), " +-", meanStats(1,2), &
        meanStats(2,1), " +-", meanStats(2,2), meanStats(3,1), " +-", meanStats(3,2)
  if (len_trim(outputFluxFile) > 0) call writeFluxASCII()
  if (computeIntensity) then   ! RadianceStats :1047-1050, :1221-1228
    allocate(RadianceStats(nx, ny, numRadDir, 2))
    off = 3 + 3*ncol + nz + ncol*nz
    do k = 1, 2
      RadianceStats(:, :, :, k) = reshape(moments(8 + (k-1)*M + off + 1 : 8 + (k-1)*M + off + ncol*numRadDir), (/ nx, ny, numRadDir /))
    end do
    RadianceStats(:, :, :, :) = solarFlux * RadianceStats(:, :, :, :) / totalNumPhotons
    RadianceStats(:, :, :, 2) = solarFlux * RadianceStats(:, :, :, 2)
    RadianceStats(:, :, :, 2) = sqrt(max(0.0_8, RadianceStats(:, :, :, 2) - RadianceStats(:, :, :, 1)**2) / (batchesCompleted - 1))
    call writeRadianceASCII()
  end if
  call finalize_Integrator(mcIntegrator)

contains
  subroutine check(what)
    character(len=*), intent(in) :: what
    if (ierr /= 0) then   ! printStatus STOPs on failure (src/userInterface_Unix.f95:19-52)
      print *, what, ": ", trim(lastMessage(mcIntegrator))
      stop 1
    end if
  end subroutine check

  subroutine momentsToStats1(s)   ! :1188-1190 for a (n, 2) array
    real(8), intent(inout) :: s(:, :)
    s(:, :) = solarFlux * s(:, :) / totalNumPhotons
    s(:, 2) = solarFlux * s(:, 2)
    s(:, 2) = sqrt(max(0.0_8, s(:, 2) - s(:, 1)**2) / (batchesCompleted - 1))
  end subroutine momentsToStats1
  subroutine momentsToStats2(s)
    real(8), intent(inout) :: s(:, :, :)
    s(:, :, :) = solarFlux * s(:, :, :) / totalNumPhotons
    s(:, :, 2) = solarFlux * s(:, :, 2)
    s(:, :, 2) = sqrt(max(0.0_8, s(:, :, 2) - s(:, :, 1)**2) / (batchesCompleted - 1))
  end subroutine momentsToStats2

  subroutine allocateDomain()
    nvox = int(nx, 8) * ny * nz
    allocate(xPosition(nx+1), yPosition(ny+1), zPosition(nz+1), totalExt(nx,ny,nz), cumExt(nx,ny,nz,nc), &
             ssa(nx,ny,nz,nc), phaseFuncI(nx,ny,nz,nc))
  end subroutine allocateDomain

  subroutine builtinDomain(name)   ! Domain-Files/i3rcStepCloud.f95:27-78, planeParallel.f95:27-78 (in km)
    character(len=*), intent(in) :: name
    real, parameter :: g = 0.85
    integer, parameter :: nLegendreCoefficients = 64
    real :: coefficients(nLegendreCoefficients)
    real(8) :: w0
    nc = 1; albedo = 0.0_8; nz = 32; ny = 1
    w0 = 0.99_8
    if (index(name, "Conservative") > 0) w0 = 1.0_8
    if (name(1:13) == "i3rcStepCloud") then
      nx = 32
      call allocateDomain()
      xPosition = 0.015625_8 * (/ (i, i = 0, nx) /); yPosition = (/ 0.0_8, 0.5_8 /)
      zPosition = 0.0078125_8 * (/ (i, i = 0, nz) /)
      totalExt(1:16, :, :) = 2.0_8 / 0.25_8; totalExt(17:32, :, :) = 18.0_8 / 0.25_8
    else if (name(1:13) == "planeParallel") then
      nx = 1
      call allocateDomain()
      xPosition = (/ 0.0_8, 0.5_8 /); yPosition = (/ 0.0_8, 0.5_8 /)
      zPosition = 0.0078125_8 * (/ (i, i = 0, nz) /)
      totalExt = 0.5_8 / 0.25_8
    else
      stop "unknown builtin domain"
    end if
    cumExt = 1.0_8; ssa = w0; phaseFuncI = 1
    coefficients = g ** (/ (i, i = 1, nLegendreCoefficients) /)
    allocate(legendreCoefficients(nLegendreCoefficients)); legendreCoefficients = coefficients
    nSteps = max(nPhaseIntervals, 9001); nEntries = 1
    allocate(table(nSteps, 1))
    call inverseTableLegendre(coefficients, table(:, 1), ierr)
    if (ierr /= 0) stop "inverseTableLegendre failed"
  end subroutine builtinDomain

  subroutine readFlatDomain(fileName)   ! mcbrat3d_amd/flatdomain.py
    character(len=*), intent(in) :: fileName
    integer :: magic
    open (unit = 3, file = fileName, access = 'stream', form = 'unformatted', status = 'old')
    read (3) magic, nx, ny, nz, nc
    if (magic /= 1296257860) stop "not a flat domain file"
    call allocateDomain()
    read (3) albedo
    read (3) xPosition, yPosition, zPosition, totalExt, cumExt, ssa, phaseFuncI
  end subroutine readFlatDomain

  subroutine loadTables()
    if (physDomainFile(1:8) == "builtin:") then
      call setInverseTable(mcIntegrator, 1, table, ierr); call check("setInverseTable")
    else
      do c = 1, nc
        read (3) nSteps, nEntries
        if (allocated(table)) deallocate(table)
        allocate(table(nSteps, nEntries))
        read (3) table
        call setInverseTable(mcIntegrator, c, table, ierr); call check("setInverseTable")
      end do
      close (3)
    end if
  end subroutine loadTables

  subroutine writeFluxASCII()   ! writeResults_ASCII, monteCarloDriver.f95:1376-1399
    open (unit = 2, file = outputFluxFile, status = 'unknown')
    write (2,'(A)') '!   I3RC Monte Carlo 3D Solar Radiative Transfer: Flux'
    write (2,'(A,A60)') '!  Property_File=', physDomainFile
    write (2,'(A,I10)')  '!  Num_Photons=', int(totalNumPhotons, 8)
    write (2,'(A,L1,A,L1)') '!  PhotonTracing=', useRayTracing, '    Russian_Roulette=', useRussianRoulette
    write (2,'(A,L1,A,F5.2)') '!  Hybrid_Phase_Func_for_Radiance=', useHybridPhaseFunsForIntenCalcs, &
                              '   Gaussian_Phase_Func_Width_deg=', hybridPhaseFunWidth
    write (2,'(A,E13.6,A,F10.7,A,F7.3)') '!  Solar_Flux=', solarFlux, '   Solar_Mu=', solarMu, '   Solar_Phi=', solarAzimuth
    write (2,'(A,F7.4)') '!  Lambertian_Surface_Albedo=', albedo
    write (2,'(A)')  '!  Output_Type= Pixel Flux'
    write (2,'(A,F7.3,A,F7.3)') '!  Upwelling_Level=', zPosition(nz+1), '   Downwelling_level=', zPosition(1)
    write (2,'(A)') '!   X      Y           Flux_Up             Flux_Down            Flux_Absorbed '
    write (2,'(A)') '!                  Mean     StdErr       Mean     StdErr       Mean     StdErr'
    write (2,'(A14,3(1X,2(1X,F9.4)))') '!  Average:   ', meanStats(1,1:2), meanStats(2,1:2), meanStats(3,1:2)
    do j = 1, ny
      do i = 1, nx
        write (2,'(2(F7.3),3(1X,2(1X,F9.4)))') sum(xPosition(i:i+1))/2., sum(yPosition(j:j+1))/2., &
              fluxUpStats(i,j,1:2), fluxDownStats(i,j,1:2), fluxAbsorbedStats(i,j,1:2)
      end do
    end do
    close (2)
  end subroutine writeFluxASCII

  subroutine writeRadianceASCII()   ! writeResults_ASCII, radiance part, monteCarloDriver.f95:1459-1494
    real :: mus(numRadDir), phis(numRadDir)
    mus = pack(intensityMus, abs(intensityMus) > 0.); phis = pack(intensityPhis, abs(intensityMus) > 0.)
    open (unit = 2, file = outputRadFile, status = 'unknown')
    write (2,'(A)') '!   I3RC Monte Carlo 3D Solar Radiative Transfer: Radiance'
    write (2,'(A,A60)') '!  Property_File=', physDomainFile
    write (2,'(A,I10)')  '!  Num_Photons=', int(totalNumPhotons, 8)
    write (2,'(A,L1,A,L1)') '!  PhotonTracing=', useRayTracing, '    Russian_Roulette=', useRussianRoulette
    write (2,'(A,L1,A,F5.2)') '!  Hybrid_Phase_Func_for_Radiance=', useHybridPhaseFunsForIntenCalcs, &
                              '   Gaussian_Phase_Func_Width_deg=', hybridPhaseFunWidth
    write (2,'(A,L1,A,F5.2)') '!  Intensity_uses_Russian_Roulette=', useRussianRouletteForIntensity, &
                              '   Intensity_Russian_Roulette_zeta_min=', zetaMin
    write (2,'(A,L1,A,F5.2)') '!  limited_intensity_contributions=', limitIntensityContributions, &
                              '   max_intensity_contribution=', maxIntensityContribution
    write (2,'(A,E13.6,A,F10.7,A,F7.3)') '!  Solar_Flux=', solarFlux, '   Solar_Mu=', solarMu, '   Solar_Phi=', solarAzimuth
    write (2,'(A,F7.4)') '!  Lambertian_Surface_Albedo=', albedo
    write (2,'(A)')  '!  Output_Type= Pixel Radiance'
    write (2,'(A,F7.3,3(A,I4))') '!  RADIANCE AT Z=', zPosition(nz+1), '   NXO=', nx, '   NYO=', ny, '   NDIR=', numRadDir
    write (2,'(A)') '!   X      Y         Radiance (Mean, StdErr)'
    do k = 1, numRadDir
      write (2,'(A,1X,F8.5,1X,F6.2,2X,A)') '! ', mus(k), phis(k), '<- (mu,phi)'
      do j = 1, ny
        do i = 1, nx
          write (2,'(2(F7.3),2(1X,F9.4))') sum(xPosition(i:i+1))/2., sum(yPosition(j:j+1))/2., RadianceStats(i,j,k,1:2)
        end do
      end do
    end do
    close (2)
  end subroutine writeRadianceASCII
end program mcbrat_driver
