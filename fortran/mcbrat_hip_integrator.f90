! mcbrat_hip_integrator.f90 -- ISO_C_BINDING shim over include/mcbrat.h.
!
! Stands where Integrators/monteCarloRadiativeTransfer.f95 stands: the same public names
! (integrator, new_Integrator, specifyParameters, computeRadiativeTransfer, reportResults,
! finalize_Integrator; reference public list :121-123), with the photon loop computeRT
! (:393-841) running in the HIP library.  The domain is handed over as the raw arrays that
! computeRT itself pulls out of type(domain) with getInfo_Domain (:434-443); INTEGRATION.md
! shows the three-line wrapper that does that inside the reference tree.
!
! Status convention: each routine returns `ierr` (0 = success) and leaves the text in
! lastMessage(); the in-tree wrapper maps it to setStateToFailure / setStateToCompleteSuccess.
module mcbrat_hip_integrator
  use, intrinsic :: iso_c_binding
  implicit none
  private

  type integrator
    private
    type(c_ptr) :: ctx = c_null_ptr
    integer     :: numX = 0, numY = 0, numZ = 0
    logical     :: readyToCompute = .false.
  end type integrator

  public :: integrator, new_Integrator, isReady_Integrator, finalize_Integrator, &
            setOpticalProperties, setInverseTable, setSolarSource, setEmissionSource, &
            specifyParameters, computeRadiativeTransfer, reportResults, &
            resetMoments, getMoments, momentsLength, lastMessage, &
            inverseTableLegendre, lastTraceMilliseconds, setAsynchronous, synchronize, &
            specifyIntensity, setForwardTable, reportIntensity, forwardTableLegendre, &
            setSurfaceDescription, setWalkOptions, setOption, getFrequencyDistr, shareMoments, chainAfter, numBadPhotons

  ! MCBRAT_ABI_VERSION of include/mcbrat.h this module was written against: mcbrat_counters has 15 fields (badPhotons) since 2
  integer(c_int), parameter :: expectedAbiVersion = 3
  interface
    function mcbrat_abi_version() bind(C, name="mcbrat_abi_version") result(v)
      import :: c_int
      integer(c_int) :: v
    end function
    function mcbrat_create(device) bind(C, name="mcbrat_create") result(ctx)
      import :: c_ptr, c_int
      integer(c_int), value :: device
      type(c_ptr) :: ctx
    end function
    subroutine mcbrat_destroy(ctx) bind(C, name="mcbrat_destroy")
      import :: c_ptr
      type(c_ptr), value :: ctx
    end subroutine
    function mcbrat_last_error(ctx) bind(C, name="mcbrat_last_error") result(msg)
      import :: c_ptr
      type(c_ptr), value :: ctx
      type(c_ptr) :: msg
    end function
    function mcbrat_set_grid(ctx, nx, ny, nz, xe, ye, ze) bind(C, name="mcbrat_set_grid") result(rc)
      import :: c_ptr, c_int, c_int32_t, c_double
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: nx, ny, nz
      real(c_double), intent(in) :: xe(*), ye(*), ze(*)
      integer(c_int) :: rc
    end function
    function mcbrat_set_optics(ctx, nc, totalExt, cumExt, ssa, pfi, albedo) bind(C, name="mcbrat_set_optics") result(rc)
      import :: c_ptr, c_int, c_int32_t, c_double
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: nc
      real(c_double), intent(in) :: totalExt(*), cumExt(*), ssa(*)
      integer(c_int32_t), intent(in) :: pfi(*)
      real(c_double), value :: albedo
      integer(c_int) :: rc
    end function
    function mcbrat_set_inverse_table(ctx, comp, nSteps, nEntries, table) bind(C, name="mcbrat_set_inverse_table") result(rc)
      import :: c_ptr, c_int, c_int32_t, c_float
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: comp, nSteps, nEntries
      real(c_float), intent(in) :: table(*)
      integer(c_int) :: rc
    end function
    function mcbrat_specify_parameters(ctx, rayTracing, roulette, lwFlag) bind(C, name="mcbrat_specify_parameters") result(rc)
      import :: c_ptr, c_int, c_int32_t, c_float
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: rayTracing, roulette
      real(c_float), value :: lwFlag
      integer(c_int) :: rc
    end function
    function mcbrat_set_surface_description(ctx, numX, numY, xPosition, yPosition, reflectance) &
        bind(C, name="mcbrat_set_surface_description") result(rc)
      import :: c_ptr, c_int, c_int32_t, c_double, c_float
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: numX, numY
      real(c_double), dimension(*), intent(in) :: xPosition, yPosition
      real(c_float), dimension(*), intent(in) :: reflectance
      integer(c_int) :: rc
    end function
    function mcbrat_set_source_solar(ctx, mu, azimuth) bind(C, name="mcbrat_set_source_solar") result(rc)
      import :: c_ptr, c_int, c_float
      type(c_ptr), value :: ctx
      real(c_float), value :: mu, azimuth
      integer(c_int) :: rc
    end function
    function mcbrat_set_source_emission(ctx, voxelWeights, fracAtms) bind(C, name="mcbrat_set_source_emission") result(rc)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: ctx
      real(c_double), intent(in) :: voxelWeights(*)
      real(c_double), value :: fracAtms
      integer(c_int) :: rc
    end function
    function mcbrat_compute_radiative_transfer(ctx, seed, firstId, ppb, nBatches, nDone) &
        bind(C, name="mcbrat_compute_radiative_transfer") result(rc)
      import :: c_ptr, c_int, c_int32_t, c_int64_t
      type(c_ptr), value :: ctx
      integer(c_int64_t), value :: seed, firstId, ppb
      integer(c_int32_t), value :: nBatches
      integer(c_int64_t), intent(out) :: nDone
      integer(c_int) :: rc
    end function
    function mcbrat_report_results(ctx, mUp, mDown, mAbs, fUp, fDown, fAbs, prof, vol) &
        bind(C, name="mcbrat_report_results") result(rc)
      import :: c_ptr, c_int, c_float
      type(c_ptr), value :: ctx
      real(c_float), intent(out) :: mUp, mDown, mAbs
      real(c_float), intent(out) :: fUp(*), fDown(*), fAbs(*), prof(*), vol(*)
      integer(c_int) :: rc
    end function
    function mcbrat_moments_length(ctx) bind(C, name="mcbrat_moments_length") result(n)
      import :: c_ptr, c_int64_t
      type(c_ptr), value :: ctx
      integer(c_int64_t) :: n
    end function
    function mcbrat_reset_moments(ctx) bind(C, name="mcbrat_reset_moments") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx
      integer(c_int) :: rc
    end function
    function mcbrat_get_moments(ctx, buf) bind(C, name="mcbrat_get_moments") result(rc)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: ctx
      real(c_double), intent(out) :: buf(*)
      integer(c_int) :: rc
    end function
    function mcbrat_specify_intensity(ctx, nDir, mus, phis, useRRI, zetaMin, useHybrid, numOrdersOrig, limitC, maxC) &
        bind(C, name="mcbrat_specify_intensity") result(rc)
      import :: c_ptr, c_int, c_int32_t, c_float
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: nDir, useRRI, useHybrid, numOrdersOrig, limitC
      real(c_float), intent(in) :: mus(*), phis(*)
      real(c_float), value :: zetaMin, maxC
      integer(c_int) :: rc
    end function
    function mcbrat_set_forward_table(ctx, component, nAngles, nEntries, table, origTable) &
        bind(C, name="mcbrat_set_forward_table") result(rc)
      import :: c_ptr, c_int, c_int32_t, c_float
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: component, nAngles, nEntries
      real(c_float), intent(in) :: table(*), origTable(*)
      integer(c_int) :: rc
    end function
    function mcbrat_report_intensity(ctx, meanIntensity, intensity) bind(C, name="mcbrat_report_intensity") result(rc)
      import :: c_ptr, c_int, c_float
      type(c_ptr), value :: ctx
      real(c_float), intent(out) :: meanIntensity(*), intensity(*)
      integer(c_int) :: rc
    end function
    function mcbrat_forward_table_legendre(nCoef, coef, nAngles, table) bind(C, name="mcbrat_forward_table_legendre") result(rc)
      import :: c_int, c_int32_t, c_float
      integer(c_int32_t), value :: nCoef, nAngles
      real(c_float), intent(in) :: coef(*)
      real(c_float), intent(out) :: table(*)
      integer(c_int) :: rc
    end function
    function mcbrat_set_async(ctx, enable) bind(C, name="mcbrat_set_async") result(rc)
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: enable
      integer(c_int) :: rc
    end function
    function mcbrat_synchronize(ctx) bind(C, name="mcbrat_synchronize") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx
      integer(c_int) :: rc
    end function
    function mcbrat_last_trace_ms(ctx) bind(C, name="mcbrat_last_trace_ms") result(ms)
      import :: c_ptr, c_float
      type(c_ptr), value :: ctx
      real(c_float) :: ms
    end function
    function mcbrat_chain_after(ctx, previous) bind(C, name="mcbrat_chain_after") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx, previous
      integer(c_int) :: rc
    end function
    function mcbrat_get_counters(ctx, counters) bind(C, name="mcbrat_get_counters") result(rc)
      import :: c_ptr, c_int, c_int64_t
      type(c_ptr), value :: ctx
      integer(c_int64_t), dimension(15), intent(out) :: counters   ! mcbrat_counters: 14 event / loop counters, then badPhotons
      integer(c_int) :: rc
    end function
    function mcbrat_set_walk_options(ctx, layerSkip, blockWalk) bind(C, name="mcbrat_set_walk_options") result(rc)
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: layerSkip, blockWalk
      integer(c_int) :: rc
    end function
    function mcbrat_set_option(ctx, name, value) bind(C, name="mcbrat_set_option") result(rc)
      import :: c_ptr, c_int, c_int32_t, c_char
      type(c_ptr), value :: ctx
      character(kind=c_char), dimension(*), intent(in) :: name
      integer(c_int32_t), value :: value
      integer(c_int) :: rc
    end function
    function mcbrat_frequency_distribution(ctx, seed, firstDraw, numLambda, cdf, totalPhotons, distribution) &
        bind(C, name="mcbrat_frequency_distribution") result(rc)
      import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
      type(c_ptr), value :: ctx
      integer(c_int64_t), value :: seed, firstDraw, totalPhotons
      integer(c_int32_t), value :: numLambda
      real(c_double), intent(in) :: cdf(*)
      integer(c_int64_t), intent(out) :: distribution(*)
      integer(c_int) :: rc
    end function
    function mcbrat_moments_device_pointer(ctx) bind(C, name="mcbrat_moments_device_pointer") result(p)
      import :: c_ptr
      type(c_ptr), value :: ctx
      type(c_ptr) :: p
    end function
    function mcbrat_bind_moments(ctx, deviceBuffer) bind(C, name="mcbrat_bind_moments") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx, deviceBuffer
      integer(c_int) :: rc
    end function
    function mcbrat_inverse_table_legendre(nCoef, coef, nSteps, table) bind(C, name="mcbrat_inverse_table_legendre") result(rc)
      import :: c_int, c_int32_t, c_float
      integer(c_int32_t), value :: nCoef, nSteps
      real(c_float), intent(in) :: coef(*)
      real(c_float), intent(out) :: table(*)
      integer(c_int) :: rc
    end function
  end interface

contains
  !------------------------------------------------------------------------------------------
  function lastMessage(this) result(msg)
    type(integrator), intent(in) :: this
    character(len=256) :: msg
    character(kind=c_char), pointer :: chars(:)
    type(c_ptr) :: p
    integer :: i
    msg = ""
    p = mcbrat_last_error(this%ctx)
    if (.not. c_associated(p)) return
    call c_f_pointer(p, chars, (/ 256 /))
    do i = 1, 256
      if (chars(i) == c_null_char) exit
      msg(i:i) = chars(i)
    end do
  end function lastMessage
  !------------------------------------------------------------------------------------------
  ! new_Integrator(atmosphere, status): here from the cell edges getInfo_Domain returns
  function new_Integrator(xPosition, yPosition, zPosition, device, ierr) result(new)
    real(8), dimension(:), intent(in) :: xPosition, yPosition, zPosition
    integer,               intent(in) :: device
    integer,               intent(out):: ierr
    type(integrator) :: new
    new%ctx = c_null_ptr
    if (mcbrat_abi_version() /= expectedAbiVersion) then
      ierr = 2   ! "new_Integrator: libmcbrat_hip.so is of another ABI version than this module" (a stale build: the
      return     !  dimension(15) counter buffer of numBadPhotons would not be what the library writes)
    end if
    new%ctx = mcbrat_create(int(device, c_int))
    if (.not. c_associated(new%ctx)) then
      ierr = 1   ! "new_Integrator: no usable HIP device" -- there is no CPU fallback
      return
    end if
    new%numX = size(xPosition) - 1; new%numY = size(yPosition) - 1; new%numZ = size(zPosition) - 1
    ierr = mcbrat_set_grid(new%ctx, int(new%numX, c_int32_t), int(new%numY, c_int32_t), int(new%numZ, c_int32_t), &
                           xPosition, yPosition, zPosition)
    new%readyToCompute = (ierr == 0)
  end function new_Integrator
  !------------------------------------------------------------------------------------------
  logical function isReady_Integrator(this)
    type(integrator), intent(in) :: this
    isReady_Integrator = this%readyToCompute
  end function isReady_Integrator
  !------------------------------------------------------------------------------------------
  subroutine finalize_Integrator(this)
    type(integrator), intent(inout) :: this
    if (c_associated(this%ctx)) call mcbrat_destroy(this%ctx)
    this%ctx = c_null_ptr
    this%readyToCompute = .false.
  end subroutine finalize_Integrator
  !------------------------------------------------------------------------------------------
  ! What computeRT pulls from the domain (getInfo_Domain :441-443)
  subroutine setOpticalProperties(this, totalExt, cumExt, ssa, phaseFuncI, albedo, ierr)
    type(integrator),            intent(inout) :: this
    real(8), dimension(:,:,:),   intent(in)    :: totalExt
    real(8), dimension(:,:,:,:), intent(in)    :: cumExt, ssa
    integer, dimension(:,:,:,:), intent(in)    :: phaseFuncI
    real(8),                     intent(in)    :: albedo
    integer,                     intent(out)   :: ierr
    ierr = mcbrat_set_optics(this%ctx, int(size(cumExt, 4), c_int32_t), totalExt, cumExt, ssa, phaseFuncI, albedo)
  end subroutine setOpticalProperties
  !------------------------------------------------------------------------------------------
  subroutine setInverseTable(this, component, values, ierr)   ! inversePhaseFuncs(component)%values
    type(integrator),     intent(inout) :: this
    integer,              intent(in)    :: component
    real, dimension(:,:), intent(in)    :: values
    integer,              intent(out)   :: ierr
    ierr = mcbrat_set_inverse_table(this%ctx, int(component, c_int32_t), int(size(values, 1), c_int32_t), &
                                    int(size(values, 2), c_int32_t), values)
  end subroutine setInverseTable
  !------------------------------------------------------------------------------------------
  ! specifyParameters(surfaceBDRF = new_SurfaceDescription(surfaceParameters, xPosition, yPosition))
  ! (monteCarloRadiativeTransfer.f95:1173-1176): reflectance = surfaceParameters(1, :, :)
  subroutine setSurfaceDescription(this, reflectance, xPosition, yPosition, ierr)
    type(integrator),      intent(inout) :: this
    real, dimension(:,:),  intent(in)    :: reflectance
    real(8), dimension(:), intent(in)    :: xPosition, yPosition
    integer,               intent(out)   :: ierr
    if (size(reflectance, 1) /= size(xPosition) - 1 .or. size(reflectance, 2) /= size(yPosition) - 1) then
      ierr = 2   ! "new_SurfaceDescription: position vector(s) are incorrect length."
      return
    end if
    ierr = mcbrat_set_surface_description(this%ctx, int(size(xPosition), c_int32_t), int(size(yPosition), c_int32_t), &
                                          xPosition, yPosition, reflectance)
  end subroutine setSurfaceDescription
  !------------------------------------------------------------------------------------------
  subroutine setSolarSource(this, solarMu, solarAzimuth, ierr)   ! new_PhotonStream, Directional
    type(integrator), intent(inout) :: this
    real,             intent(in)    :: solarMu, solarAzimuth
    integer,          intent(out)   :: ierr
    ierr = mcbrat_set_source_solar(this%ctx, solarMu, solarAzimuth)
  end subroutine setSolarSource
  !------------------------------------------------------------------------------------------
  subroutine setEmissionSource(this, voxelWeights, fracAtmsPower, ierr)   ! new_PhotonStream, BBEmission
    type(integrator),          intent(inout) :: this
    real(8), dimension(:,:,:), intent(in)    :: voxelWeights
    real(8),                   intent(in)    :: fracAtmsPower
    integer,                   intent(out)   :: ierr
    ierr = mcbrat_set_source_emission(this%ctx, voxelWeights, fracAtmsPower)
  end subroutine setEmissionSource
  !------------------------------------------------------------------------------------------
  ! specifyParameters: the keywords the driver passes (monteCarloDriver.f95:540-572)
  subroutine specifyParameters(this, useRayTracing, useRussianRoulette, LW_flag, ierr)
    type(integrator), intent(inout) :: this
    logical,          intent(in)    :: useRayTracing, useRussianRoulette
    real,             intent(in)    :: LW_flag
    integer,          intent(out)   :: ierr
    ierr = mcbrat_specify_parameters(this%ctx, merge(1_c_int32_t, 0_c_int32_t, useRayTracing), &
                                     merge(1_c_int32_t, 0_c_int32_t, useRussianRoulette), LW_flag)
  end subroutine specifyParameters
  !------------------------------------------------------------------------------------------
  ! computeRadiativeTransfer(thisIntegrator, thisDomain, randomNumbers, incomingPhotons,
  !                          numPhotonsPerBatch, numPhotonsProcessed, status)
  ! randomNumbers -> (seed, firstPhotonId): counter-based generator keyed by photon id.
  subroutine computeRadiativeTransfer(this, seed, firstPhotonId, numPhotonsPerBatch, numBatches, &
                                      numPhotonsProcessed, ierr)
    type(integrator), intent(inout) :: this
    integer(8),       intent(in)    :: seed, firstPhotonId, numPhotonsPerBatch
    integer,          intent(in)    :: numBatches
    integer(8),       intent(out)   :: numPhotonsProcessed
    integer,          intent(out)   :: ierr
    if (.not. this%readyToCompute) then
      ierr = 1; numPhotonsProcessed = 0
      return
    end if
    ierr = mcbrat_compute_radiative_transfer(this%ctx, seed, firstPhotonId, numPhotonsPerBatch, &
                                             int(numBatches, c_int32_t), numPhotonsProcessed)
  end subroutine computeRadiativeTransfer
  !------------------------------------------------------------------------------------------
  subroutine reportResults(this, meanFluxUp, meanFluxDown, meanFluxAbsorbed, fluxUp, fluxDown, fluxAbsorbed, &
                           absorbedProfile, volumeAbsorption, ierr)
    type(integrator),         intent(in)  :: this
    real,                     intent(out) :: meanFluxUp, meanFluxDown, meanFluxAbsorbed
    real, dimension(:,:),     intent(out) :: fluxUp, fluxDown, fluxAbsorbed
    real, dimension(:),       intent(out) :: absorbedProfile
    real, dimension(:,:,:),   intent(out) :: volumeAbsorption
    integer,                  intent(out) :: ierr
    if (any(shape(fluxUp) /= (/ this%numX, this%numY /)) .or. size(absorbedProfile) /= this%numZ .or. &
        any(shape(volumeAbsorption) /= (/ this%numX, this%numY, this%numZ /))) then
      ierr = 2   ! "reportResults: ... array is the wrong size"
      return
    end if
    ierr = mcbrat_report_results(this%ctx, meanFluxUp, meanFluxDown, meanFluxAbsorbed, fluxUp, fluxDown, fluxAbsorbed, &
                                 absorbedProfile, volumeAbsorption)
  end subroutine reportResults
  !------------------------------------------------------------------------------------------
  integer(8) function momentsLength(this)
    type(integrator), intent(in) :: this
    momentsLength = mcbrat_moments_length(this%ctx)
  end function momentsLength
  subroutine resetMoments(this, ierr)
    type(integrator), intent(inout) :: this
    integer, intent(out) :: ierr
    ierr = mcbrat_reset_moments(this%ctx)
  end subroutine resetMoments
  subroutine getMoments(this, buffer, ierr)   ! header(8) + S1(M) + S2(M), see include/mcbrat.h
    type(integrator), intent(inout) :: this
    real(8), dimension(:), intent(out) :: buffer
    integer, intent(out) :: ierr
    ierr = mcbrat_get_moments(this%ctx, buffer)
  end subroutine getMoments
  ! specifyParameters, intensity keywords (:1046-1292): directions and variance-reduction choices
  subroutine specifyIntensity(this, intensityMus, intensityPhis, useRussianRouletteForIntensity, zetaMin, &
                              useHybridPhaseFunsForIntenCalcs, numOrdersOrigPhaseFunIntenCalcs, &
                              limitIntensityContributions, maxIntensityContribution, ierr)
    type(integrator),   intent(inout) :: this
    real, dimension(:), intent(in)    :: intensityMus, intensityPhis
    logical,            intent(in)    :: useRussianRouletteForIntensity, useHybridPhaseFunsForIntenCalcs, &
                                         limitIntensityContributions
    real,               intent(in)    :: zetaMin, maxIntensityContribution
    integer,            intent(in)    :: numOrdersOrigPhaseFunIntenCalcs
    integer,            intent(out)   :: ierr
    ierr = mcbrat_specify_intensity(this%ctx, int(size(intensityMus), c_int32_t), intensityMus, intensityPhis, &
                                    merge(1_c_int32_t, 0_c_int32_t, useRussianRouletteForIntensity), zetaMin, &
                                    merge(1_c_int32_t, 0_c_int32_t, useHybridPhaseFunsForIntenCalcs), &
                                    int(numOrdersOrigPhaseFunIntenCalcs, c_int32_t), &
                                    merge(1_c_int32_t, 0_c_int32_t, limitIntensityContributions), maxIntensityContribution)
  end subroutine specifyIntensity
  ! tabulatedPhaseFunctions(component)%values / tabulatedOrigPhaseFunctions(component)%values (nAngles, nEntries)
  subroutine setForwardTable(this, component, values, origValues, ierr)
    type(integrator),     intent(inout) :: this
    integer,              intent(in)    :: component
    real, dimension(:,:), intent(in)    :: values, origValues
    integer,              intent(out)   :: ierr
    ierr = mcbrat_set_forward_table(this%ctx, int(component, c_int32_t), int(size(values, 1), c_int32_t), &
                                    int(size(values, 2), c_int32_t), values, origValues)
  end subroutine setForwardTable
  subroutine reportIntensity(this, meanIntensity, intensity, ierr)   ! reportResults(meanIntensity, intensity) :980-1010
    type(integrator),       intent(in)  :: this
    real, dimension(:),     intent(out) :: meanIntensity
    real, dimension(:,:,:), intent(out) :: intensity
    integer,                intent(out) :: ierr
    if (size(intensity, 1) /= this%numX .or. size(intensity, 2) /= this%numY .or. size(intensity, 3) /= size(meanIntensity)) then
      ierr = 1   ! "reportResults: intensity array has wrong dimensions."
      return
    end if
    ierr = mcbrat_report_intensity(this%ctx, meanIntensity, intensity)
  end subroutine reportIntensity
  subroutine forwardTableLegendre(coefficients, table, ierr)   ! tabulateForwardPhaseFunctions for one Legendre entry
    real, dimension(:), intent(in)  :: coefficients
    real, dimension(:), intent(out) :: table
    integer,            intent(out) :: ierr
    ierr = mcbrat_forward_table_legendre(int(size(coefficients), c_int32_t), coefficients, int(size(table), c_int32_t), table)
  end subroutine forwardTableLegendre
  ! Per-batch callers (the driver's loop, monteCarloDriver.f95:1008): let consecutive calls overlap on the GPU;
  ! reportResults / getMoments synchronise by themselves.
  subroutine setAsynchronous(this, enable, ierr)
    type(integrator), intent(inout) :: this
    logical, intent(in) :: enable
    integer, intent(out) :: ierr
    ierr = mcbrat_set_async(this%ctx, merge(1_c_int32_t, 0_c_int32_t, enable))
  end subroutine setAsynchronous
  subroutine synchronize(this, ierr)
    type(integrator), intent(inout) :: this
    integer, intent(out) :: ierr
    ierr = mcbrat_synchronize(this%ctx)
  end subroutine synchronize
  !------------------------------------------------------------------------------------------
  ! layerSkip (one-extinction layers and the clear-air flight) / blockWalk: .false. restores the reference's face-by-face walk (include/mcbrat.h)
  subroutine setWalkOptions(this, layerSkip, blockWalk, ierr)
    type(integrator), intent(inout) :: this
    logical, intent(in) :: layerSkip, blockWalk
    integer, intent(out) :: ierr
    ierr = mcbrat_set_walk_options(this%ctx, merge(1_c_int32_t, 0_c_int32_t, layerSkip), merge(1_c_int32_t, 0_c_int32_t, blockWalk))
  end subroutine setWalkOptions
  ! Scheduling options by name (include/mcbrat.h: mcbrat_set_option), e.g. setOption(this, "jumpThreshold", 16, ierr); none of them
  ! changes a result.  The reference has no counterpart.
  subroutine setOption(this, name, value, ierr)
    type(integrator), intent(inout) :: this
    character(len=*), intent(in) :: name
    integer, intent(in) :: value
    integer, intent(out) :: ierr
    ierr = mcbrat_set_option(this%ctx, trim(name) // c_null_char, int(value, c_int32_t))
  end subroutine setOption
  ! computeRT's nBad (Integrators/monteCarloRadiativeTransfer.f95:420, :562-563: photons dropped because their step was not
  ! positive): here the photons dropped because a loop bound of the kernels was reached, since this integrator was created
  ! (mcbrat_counters.badPhotons, include/mcbrat.h).  Synchronises.
  integer(8) function numBadPhotons(this)
    type(integrator), intent(in) :: this
    integer(c_int64_t), dimension(15) :: counters
    numBadPhotons = -1
    if (mcbrat_get_counters(this%ctx, counters) == 0) numBadPhotons = counters(15)
  end function numBadPhotons
  ! getFrequencyDistr (src/emissionAndBroadBandWeights.f95:552-572) with the draws made and counted on the device
  subroutine getFrequencyDistr(this, CDF, totalPhotons, iseed, distribution, ierr)
    type(integrator), intent(inout) :: this
    real(8), dimension(:), intent(in) :: CDF
    integer(8), intent(in) :: totalPhotons
    integer, intent(in) :: iseed
    integer(8), dimension(:), intent(out) :: distribution
    integer, intent(out) :: ierr
    ierr = mcbrat_frequency_distribution(this%ctx, int(iseed, c_int64_t), 0_c_int64_t, int(size(CDF), c_int32_t), CDF, &
                                         int(totalPhotons, c_int64_t), distribution)
  end subroutine getFrequencyDistr
  ! spectrally integrated runs: `this` (another wavelength's integrator on the same grid) accumulates into `first`'s moments
  subroutine shareMoments(this, first, ierr)
    type(integrator), intent(inout) :: this, first
    integer, intent(out) :: ierr
    ierr = mcbrat_bind_moments(this%ctx, mcbrat_moments_device_pointer(first%ctx))
  end subroutine shareMoments
  ! ... and, with both integrators asynchronous (setAsynchronous), `this` folds the batches of its NEXT call after everything
  ! `previous` has enqueued so far, while the two integrators' tracing kernels overlap (mcbrat_chain_after, include/mcbrat.h)
  subroutine chainAfter(this, previous, ierr)
    type(integrator), intent(inout) :: this, previous
    integer, intent(out) :: ierr
    ierr = mcbrat_chain_after(this%ctx, previous%ctx)
  end subroutine chainAfter
  real function lastTraceMilliseconds(this)
    type(integrator), intent(in) :: this
    lastTraceMilliseconds = mcbrat_last_trace_ms(this%ctx)
  end function lastTraceMilliseconds
  !------------------------------------------------------------------------------------------
  subroutine inverseTableLegendre(coefficients, table, ierr)   ! computeInversePhaseFunction
    real, dimension(:), intent(in)  :: coefficients
    real, dimension(:), intent(out) :: table
    integer,            intent(out) :: ierr
    ierr = mcbrat_inverse_table_legendre(int(size(coefficients), c_int32_t), coefficients, &
                                         int(size(table), c_int32_t), table)
  end subroutine inverseTableLegendre
end module mcbrat_hip_integrator
