! greb_host.f90 -- thin Fortran host of the MI355X GREB engine.
!
! Keeps the reference's user-facing conventions and hands the time loops to the C-ABI engine
! (include/greb_engine.h) through iso_c_binding:
!   * command line: ./greb_host [namelist]            (reference: src/greb.f90:1032-1038)
!   * namelist groups physics_par / numerics_par / diagnostics_par / co2_par with the
!     reference's names and defaults                  (src/greb.f90:49-55,68-104,128-134,152-156;
!                                                       doc/namelist.md)
!   * the ten raw fp32 direct-access files under input/ (src/greb.f90:1018-1027,1073-1085)
!   * co2_ppm padding rule                             (src/greb.f90:1053-1061)
!   * output file name <output_file>[_<ens_id>], 5 direct-access records per month in the order
!     Tsurf, Tair, Tocean, q, albedo                   (src/greb.f90:1064-1068,174,978-982)
!   * the console trace                                (src/greb.f90:219,224-225,941,954,1070)
! Control crosses into the engine twice per run (flux-correction phase, scenario phase); the
! derived fields of greb_model's preamble are computed by the engine's create().
! An optional fifth group &ENGINE_PAR (strict, device, corr_file, nx, ny, chunk_years) selects reference-order
! arithmetic, the GPU, a flux-correction cache file, the grid (the reference's is compile-time, src/greb.f90:36) and
! how many scenario years are taken per engine call (default: as many as fit a 2 GB host buffer).
! An optional sixth group &ENSEMBLE_PAR runs an ENSEMBLE in one engine: in the reference an ensemble is N
! processes with N namelists that differ in ens_id (src/greb.f90:153,1064-1068); here the N members share one
! greb_engine_create call (the GPU integrates them side by side) and each member's monthly means go to its own
! <output_file>_<ens_id> in the reference's record layout, so existing per-ens_id analysis scripts run unchanged:
!   n_members            number of members (default 1 = the reference's single run)
!   ens_ids(:)           their ids (default '001', '002', ...; with one member the &DIAGNOSTICS_PAR ens_id)
!   co2_levels(:)        constant CO2 [ppm] per member, or
!   co2_lo, co2_hi       a linear sweep over the members (BASELINE config 4: 280 .. 1120); neither given: every
!                        member follows the &CO2_PAR series
!   ens_da_ice(:), ens_a_no_ice(:), ens_a_cloud(:), ens_kappa(:)
!                        per-member values of these &PHYSICS_PAR parameters (BASELINE config 5); unset = the
!                        &PHYSICS_PAR value.  Members with their own physics get their own flux correction.
!   n_procs, proc_id     ONE ensemble over SEVERAL host processes (one per GPU: the reference's own convention of N
!                        processes with N ens_ids, src/greb.f90:153,1064-1068): the namelist describes the whole
!                        ensemble, process proc_id (0-based) of n_procs integrates its contiguous block of members --
!                        block sizes differ by at most one, the rule of greb_climate_model_amd/ensemble.py:partition --
!                        and writes only their <output_file>_<ens_id> files.  Also settable from the command line, so
!                        that all processes share one namelist file:   greb_host <namelist> <proc_id> <n_procs> [plan]   (or: greb_host <namelist> plan)
!                        (`plan` prints the block and stops before any input is read or any GPU is touched).  With
!                        &ENGINE_PAR device unset, process proc_id uses GPU proc_id.  tools/launch_ensemble.py starts them.
! &ENGINE_PAR one_launch (384- / 192-wide grids): 1 = the circulation call as one launch (GREB_F_PERSISTENT), 0 = one launch
!                        per sub-step (GREB_F_NO_PERSISTENT), unset = the engine times both and keeps the faster -- except
!                        where several host processes were given the same device: there 0, see below.
! The interface module is host/greb_c_api.f90.

program greb_host
  use iso_c_binding
  use greb_c_api
  implicit none

  integer, parameter :: nstep = 730, max_members = 4096
  real, parameter :: unset = -999.
  integer :: nx, ny
  ! ---- namelist variables under the reference's names
  real :: pi, sig, rho_ocean, rho_land, rho_air, cp_ocean, cp_land, cp_air, eps
  real :: d_ocean, d_land, d_air, ct_sens, da_ice, a_no_ice, a_cloud
  real :: Tl_ice1, Tl_ice2, To_ice1, To_ice2
  real :: co_turb, kappa, ce, cq_latent, cq_rain, z_air, z_vapor, r_qviwv
  real :: p_emi(10)
  real :: co2_flux
  real, allocatable :: co2_ppm(:)
  integer :: ipx, ipy, time_flux, time_scnr, year0
  character(len=120) :: output_file
  character(len=10)  :: ens_id
  logical :: strict, have_cache
  integer :: device
  character(len=200) :: corr_file
  real(c_float), allocatable :: corr(:), state5(:)
  namelist / physics_par / pi, sig, rho_ocean, rho_land, rho_air, cp_ocean, cp_land, cp_air, eps, &
       d_ocean, d_land, d_air, ct_sens, da_ice, a_no_ice, a_cloud, Tl_ice1, Tl_ice2, To_ice1, To_ice2, &
       co_turb, kappa, ce, cq_latent, cq_rain, z_air, z_vapor, r_qviwv, p_emi
  namelist / numerics_par / ipx, ipy, time_flux, time_scnr, year0
  namelist / diagnostics_par / output_file, ens_id
  namelist / co2_par / co2_ppm, co2_flux
  integer :: chunk_years, one_launch
  logical :: device_given
  namelist / engine_par / strict, device, corr_file, nx, ny, chunk_years, one_launch
  ! ---- ensemble
  integer :: n_members, n_total, n_procs, proc_id, first_member, gm
  logical :: plan_only
  character(len=16) :: arg
  character(len=10), allocatable :: ens_ids(:)
  real, allocatable :: co2_levels(:), ens_da_ice(:), ens_a_no_ice(:), ens_a_cloud(:), ens_kappa(:)
  real :: co2_lo, co2_hi
  namelist / ensemble_par / n_members, ens_ids, co2_levels, co2_lo, co2_hi, ens_da_ice, ens_a_no_ice, ens_a_cloud, ens_kappa, &
       n_procs, proc_id
  type(greb_member_overrides), allocatable, target :: ov(:)
  real(c_float), allocatable :: co2_all(:)
  integer, allocatable :: units(:)
  logical :: own_physics
  integer :: m, y0, cy, chunk
  integer(8) :: rec_year, moff

  type(greb_params) :: prm
  type(greb_fields) :: fld
  type(c_ptr) :: eng
  real(c_float), allocatable, target :: z_topo(:,:), glacier(:,:), sw_solar(:,:)
  real(c_float), allocatable, target :: tclim(:,:,:), qclim(:,:,:), uclim(:,:,:), vclim(:,:,:), &
       mldclim(:,:,:), cldclim(:,:,:), swetclim(:,:,:)
  real(c_float), allocatable :: monthly(:), yearly(:), yflux(:)
  character(len=256) :: nml_file
  character(len=131) :: out_full
  integer :: n, i, irec, nrec, ios, nargs
  integer(c_int) :: rc, flags
  integer(8) :: off
  real :: year

  ! ---- defaults: the engine's copy of the reference defaults, then the namelist on top
  call greb_params_default(prm)
  pi = prm%pi; sig = prm%sig; rho_ocean = prm%rho_ocean; rho_land = prm%rho_land; rho_air = prm%rho_air
  cp_ocean = prm%cp_ocean; cp_land = prm%cp_land; cp_air = prm%cp_air; eps = prm%eps
  d_ocean = prm%d_ocean; d_land = prm%d_land; d_air = prm%d_air; ct_sens = prm%ct_sens
  da_ice = prm%da_ice; a_no_ice = prm%a_no_ice; a_cloud = prm%a_cloud
  Tl_ice1 = prm%Tl_ice1; Tl_ice2 = prm%Tl_ice2; To_ice1 = prm%To_ice1; To_ice2 = prm%To_ice2
  co_turb = prm%co_turb; kappa = prm%kappa; ce = prm%ce; cq_latent = prm%cq_latent; cq_rain = prm%cq_rain
  z_air = prm%z_air; z_vapor = prm%z_vapor; r_qviwv = prm%r_qviwv; p_emi = prm%p_emi
  co2_flux = prm%co2_flux
  ipx = 1; ipy = 1; time_flux = 0; time_scnr = 0; year0 = 1940
  output_file = 'output/scenario'; ens_id = ''
  strict = .false.; device = -1; corr_file = ''; nx = 96; ny = 48; chunk_years = 0; one_launch = -1
  n_members = 1; co2_lo = unset; co2_hi = unset; n_procs = 1; proc_id = 0; plan_only = .false.
  allocate(ens_ids(max_members), co2_levels(max_members), ens_da_ice(max_members), ens_a_no_ice(max_members), &
       ens_a_cloud(max_members), ens_kappa(max_members))
  ens_ids = ''; co2_levels = unset; ens_da_ice = unset; ens_a_no_ice = unset; ens_a_cloud = unset; ens_kappa = unset

  nargs = command_argument_count()
  nml_file = 'namelist'
  if (nargs >= 1) call get_command_argument(1, nml_file)
  open(10, file=trim(nml_file), action='read', status='old')
  read(10, nml=physics_par)
  read(10, nml=numerics_par)
  read(10, nml=diagnostics_par)
  allocate(co2_ppm(max(time_scnr, 1)))
  co2_ppm = -1.
  read(10, nml=co2_par)
  read(10, nml=engine_par, iostat=ios)   ! optional group
  rewind(10)
  read(10, nml=ensemble_par, iostat=ios) ! optional group
  close(10)
  if (n_members < 1 .or. n_members > max_members) then
     print*, 'greb_host: n_members must be 1 ..', max_members
     error stop 1
  end if
  ! ---- this process's block of the ensemble
  if (nargs >= 3) then
     call get_command_argument(2, arg); read(arg, *, iostat=ios) proc_id
     if (ios == 0) then
        call get_command_argument(3, arg); read(arg, *, iostat=ios) n_procs
     end if
     if (ios /= 0) then
        print*, 'greb_host: usage: greb_host [namelist [proc_id n_procs [plan]]]'
        error stop 1
     end if
  end if
  if (nargs >= 4) then
     call get_command_argument(4, arg)
     plan_only = trim(arg) == 'plan'
  else if (nargs == 2) then  ! greb_host <namelist> plan: the block from the namelist's own n_procs / proc_id
     call get_command_argument(2, arg)
     plan_only = trim(arg) == 'plan'
  end if
  if (n_procs < 1 .or. proc_id < 0 .or. proc_id >= n_procs) then
     print*, 'greb_host: need 0 <= proc_id < n_procs, got ', proc_id, n_procs
     error stop 1
  end if
  n_total = n_members
  first_member = proc_id*(n_total/n_procs) + min(proc_id, mod(n_total, n_procs)) + 1
  n_members = n_total/n_procs
  if (proc_id < mod(n_total, n_procs)) n_members = n_members + 1
  device_given = device >= 0
  if (device < 0) device = merge(proc_id, 0, n_procs > 1)
  if (n_procs > 1) print '(a,i0,a,i0,a,i0,a,i0,a,i0,a,i0)', ' % ENSEMBLE BLOCK; process ', proc_id, ' of ', n_procs, &
       ': members ', first_member, ' .. ', first_member + n_members - 1, ' of ', n_total, ' on device ', device

  ! a series shorter than the run is continued with its last value; none at all means 2xCO2
  if (co2_ppm(1) == -1.) co2_ppm(1) = 680.
  do i = 2, time_scnr
     if (co2_ppm(i) < 0.) co2_ppm(i) = co2_ppm(i-1)
  end do

  if (len_trim(ens_id) == 0) then
     out_full = trim(output_file)
  else
     out_full = trim(output_file) // '_' // trim(ens_id)
  end if

  print*,'% diagonstic point lat/lon: ',(180./ny)*ipy-90, (360./nx)*ipx

  ! ---- the ensemble: ids, CO2 series [years, member] and physics overrides per member
  do m = 1, n_total
     if (len_trim(ens_ids(m)) == 0) then
        if (n_total == 1) then
           ens_ids(m) = ens_id
        else
           write(ens_ids(m), '(i3.3)') m
        end if
     end if
  end do
  ! from here on m = 1 .. n_members counts THIS process's members; member m is member first_member + m - 1 of the ensemble
  do m = 1, n_members
     gm = first_member + m - 1
     ens_ids(m) = ens_ids(gm); co2_levels(m) = co2_levels(gm)
     ens_da_ice(m) = ens_da_ice(gm); ens_a_no_ice(m) = ens_a_no_ice(gm); ens_a_cloud(m) = ens_a_cloud(gm); ens_kappa(m) = ens_kappa(gm)
  end do
  if (plan_only) then
     do m = 1, n_members
        print '(a,i0,a,a)', ' % member ', first_member + m - 1, ' ens_id ', trim(ens_ids(m))
     end do
     stop
  end if
  if (n_members == 0) then
     print*, '% no member of the ensemble falls to this process'
     stop
  end if
  allocate(co2_all(max(time_scnr,1)*n_members))
  do m = 1, n_members
     gm = first_member + m - 1
     do i = 1, time_scnr
        co2_all((m-1)*time_scnr + i) = co2_ppm(i)
        if (co2_lo /= unset .and. co2_hi /= unset .and. n_total > 1) &
             co2_all((m-1)*time_scnr + i) = real(dble(co2_lo) + dble(co2_hi - co2_lo)*dble(gm-1)/dble(n_total-1))
        if (co2_levels(m) /= unset) co2_all((m-1)*time_scnr + i) = co2_levels(m)
     end do
  end do
  allocate(ov(n_members))
  own_physics = .false.
  do m = 1, n_members
     ov(m)%da_ice = pick(ens_da_ice(m)); ov(m)%a_no_ice = pick(ens_a_no_ice(m))
     ov(m)%a_cloud = pick(ens_a_cloud(m)); ov(m)%kappa = pick(ens_kappa(m))
     if (ens_da_ice(m) /= unset .or. ens_a_no_ice(m) /= unset .or. ens_a_cloud(m) /= unset .or. ens_kappa(m) /= unset) &
          own_physics = .true.
  end do
  if (own_physics .and. len_trim(corr_file) > 0) then
     print*, 'greb_host: corr_file caches ONE flux correction; members with their own physics each have their own'
     error stop 1
  end if

  ! ---- boundary data
  allocate(z_topo(nx,ny), glacier(nx,ny), sw_solar(ny,nstep))
  allocate(tclim(nx,ny,nstep), qclim(nx,ny,nstep), uclim(nx,ny,nstep), vclim(nx,ny,nstep))
  allocate(mldclim(nx,ny,nstep), cldclim(nx,ny,nstep), swetclim(nx,ny,nstep))
  call read_records('input/topography', z_topo, 1)
  call read_records('input/glacier.masks', glacier, 1)
  open(15, file='input/solar.radiation', access='direct', form='unformatted', recl=4*ny*nstep, status='old')
  read(15, rec=1) sw_solar
  close(15)
  call read_records('input/tsurf', tclim, nstep)
  call read_records('input/vapor', qclim, nstep)
  call read_records('input/soil.moisture', swetclim, nstep)
  call read_records('input/zonal.wind', uclim, nstep)
  call read_records('input/meridional.wind', vclim, nstep)
  call read_records('input/ocean.mld', mldclim, nstep)
  call read_records('input/cloud.cover', cldclim, nstep)

  ! ---- hand everything to the engine
  prm%pi = pi; prm%sig = sig; prm%rho_ocean = rho_ocean; prm%rho_land = rho_land; prm%rho_air = rho_air
  prm%cp_ocean = cp_ocean; prm%cp_land = cp_land; prm%cp_air = cp_air; prm%eps = eps
  prm%d_ocean = d_ocean; prm%d_land = d_land; prm%d_air = d_air; prm%ct_sens = ct_sens
  prm%da_ice = da_ice; prm%a_no_ice = a_no_ice; prm%a_cloud = a_cloud
  prm%Tl_ice1 = Tl_ice1; prm%Tl_ice2 = Tl_ice2; prm%To_ice1 = To_ice1; prm%To_ice2 = To_ice2
  prm%co_turb = co_turb; prm%kappa = kappa; prm%ce = ce; prm%cq_latent = cq_latent; prm%cq_rain = cq_rain
  prm%z_air = z_air; prm%z_vapor = z_vapor; prm%r_qviwv = r_qviwv; prm%p_emi = p_emi
  prm%co2_flux = co2_flux
  prm%ipx = ipx; prm%ipy = ipy; prm%year0 = year0
  fld%z_topo = c_loc(z_topo); fld%glacier = c_loc(glacier); fld%sw_solar = c_loc(sw_solar)
  fld%tclim = c_loc(tclim); fld%qclim = c_loc(qclim); fld%uclim = c_loc(uclim); fld%vclim = c_loc(vclim)
  fld%mldclim = c_loc(mldclim); fld%cldclim = c_loc(cldclim); fld%swetclim = c_loc(swetclim)
  flags = 0
  if (strict) flags = 1
  ! 384- / 192-wide grids: the circulation call as ONE launch waits inside the kernel for its own strips, which must all
  ! be resident at once; engines of one process share the device through a ledger, PROCESSES do not see each other.  So
  ! several host processes told to use the SAME device (&ENGINE_PAR device given with n_procs > 1) take one launch per
  ! sub-step unless the namelist says otherwise; one process per GPU (the default placement) lets the engine choose.
  if (one_launch < 0 .and. n_procs > 1 .and. device_given) one_launch = 0
  if (one_launch == 0) flags = flags + 8    ! GREB_F_NO_PERSISTENT
  if (one_launch > 0) flags = flags + 16    ! GREB_F_PERSISTENT
  eng = c_null_ptr
  if (own_physics) then
     rc = greb_engine_create(prm, int(nx, c_int), int(ny, c_int), fld, int(n_members, c_int), c_loc(ov), int(device, c_int), flags, eng)
  else
     rc = greb_engine_create(prm, int(nx, c_int), int(ny, c_int), fld, int(n_members, c_int), c_null_ptr, int(device, c_int), flags, eng)
  end if
  call engine_check(rc, eng, 'greb_engine_create')

  ! Flux-correction cache (SURVEY.md 8f-2): the reference recomputes qflux_correction
  ! (src/greb.f90:311-364) in every run.  With corr_file set, the phase's products -- TF/qF/ToF_correct,
  ! cap_surf and the end state -- are written after the first run and read back by later ones, which
  ! then start the scenario directly (same records as every other GREB file: nx*ny fp32 each).
  have_cache = .false.
  if (len_trim(corr_file) > 0) inquire(file=trim(corr_file), exist=have_cache)
  if (len_trim(corr_file) > 0) allocate(corr(3*nstep*nx*ny), state5(5*nx*ny))
  if (have_cache) then
     print*,'% FLUX CORRECTION read from ', trim(corr_file)
     open(23, file=trim(corr_file), access='direct', form='unformatted', recl=4*nx*ny, status='old')
     do irec = 1, 3*nstep
        read(23, rec=irec) corr((irec-1)*nx*ny+1 : irec*nx*ny)
     end do
     do irec = 1, 5
        read(23, rec=3*nstep+irec) state5((irec-1)*nx*ny+1 : irec*nx*ny)
     end do
     close(23)
     rc = greb_engine_set_corrections(eng, -1_c_int, corr, state5)
     call engine_check(rc, eng, 'greb_engine_set_corrections')
  else
     print*,'% FLUX CORRECTION RUN; years = ', time_flux, ' co2 = ', co2_flux
     allocate(yflux(2*max(time_flux,1)*n_members))
     rc = greb_engine_flux_correction(eng, int(time_flux, c_int), yflux)
     call engine_check(rc, eng, 'greb_engine_flux_correction')
     do m = 1, n_members
        if (m > 1 .and. .not. own_physics) exit   ! one shared flux correction: one trace
        if (own_physics) print*, '% MEMBER ', trim(ens_ids(m))
        if (time_flux > 0) print *, 'console output: year, co2, global avg temp, avg temp for ipx/ipy'
        do n = 1, time_flux
           print *, 0.0, co2_flux, yflux(2*((m-1)*time_flux + n)-1), yflux(2*((m-1)*time_flux + n))
        end do
     end do
     if (len_trim(corr_file) > 0) then
        rc = greb_engine_get_corrections(eng, 0_c_int, corr, state5)
        call engine_check(rc, eng, 'greb_engine_get_corrections')
        open(23, file=trim(corr_file), access='direct', form='unformatted', recl=4*nx*ny)
        do irec = 1, 3*nstep
           write(23, rec=irec) corr((irec-1)*nx*ny+1 : irec*nx*ny)
        end do
        do irec = 1, 5
           write(23, rec=3*nstep+irec) state5((irec-1)*nx*ny+1 : irec*nx*ny)
        end do
        close(23)
        print*,'% FLUX CORRECTION saved in ', trim(corr_file)
     end if
  end if

  print*,'% MODEL RUN; years = ', time_scnr
  if (n_total == 1) print*,'% saving output in file ', out_full
  if (n_total > 1) print*,'% saving output in files ', trim(output_file), '_<ens_id>; members = ', n_members
  if (time_scnr > 0) then
     ! the engine's clock continues across calls (include/greb_engine.h), so a long run is taken in chunks of whole
     ! years that keep the monthly-mean buffer [member][year][12][5][ny][nx] below ~2 GB
     rec_year = int(12*5, 8)*nx*ny
     chunk = int(max(1_8, min(int(time_scnr, 8), 500000000_8/(rec_year*n_members))))
     if (chunk_years > 0) chunk = min(chunk_years, time_scnr)   ! &ENGINE_PAR chunk_years: a smaller host buffer
     allocate(monthly(rec_year*chunk*n_members), yearly(2*time_scnr*n_members), units(n_members))
     do m = 1, n_members
        if (n_total == 1) then
           open(newunit=units(m), file=out_full, access='direct', form='unformatted', recl=4*nx*ny)
        else
           open(newunit=units(m), file=trim(output_file)//'_'//trim(ens_ids(m)), access='direct', form='unformatted', recl=4*nx*ny)
        end if
     end do
     do y0 = 0, time_scnr - 1, chunk
        cy = min(chunk, time_scnr - y0)
        rc = greb_engine_run(eng, int(cy, c_int), chunk_co2(y0, cy), monthly, yearly(2*y0*n_members+1:), 0_c_int)
        call engine_check(rc, eng, 'greb_engine_run')
        do m = 1, n_members
           moff = int(m-1, 8)*cy*rec_year
           do irec = 1, cy*60
              off = moff + int(irec-1, 8)*nx*ny
              write(units(m), rec=y0*60 + irec) monthly(off+1:off+nx*ny)   ! :978-982
           end do
        end do
     end do
     do m = 1, n_members
        close(units(m))
        if (n_total > 1) print*, '% MEMBER ', trim(ens_ids(m))
        print *, 'console output: year, co2, global avg temp, avg temp for ipx/ipy'
        year = year0
        do n = 1, time_scnr
           ! yearly holds one [member][cy][2] block per chunk
           y0 = ((n-1)/chunk)*chunk; cy = min(chunk, time_scnr - y0)
           i = 2*y0*n_members + 2*((m-1)*cy + (n-1-y0))
           print *, year, co2_all((m-1)*time_scnr + n), yearly(i+1), yearly(i+2)
           year = year + 1
        end do
     end do
  end if
  rc = greb_engine_destroy(eng)

contains
  real(c_float) function pick(v)
    use ieee_arithmetic
    real, intent(in) :: v
    pick = v
    if (v == unset) pick = ieee_value(1.0_c_float, ieee_quiet_nan)   ! NaN = keep the engine-wide value
  end function
  function chunk_co2(y0, cy) result(c)
    ! [member][cy] slice of the [member][time_scnr] series
    integer, intent(in) :: y0, cy
    real(c_float) :: c(cy*n_members)
    integer :: mm
    do mm = 1, n_members
       c((mm-1)*cy+1 : mm*cy) = co2_all((mm-1)*time_scnr + y0 + 1 : (mm-1)*time_scnr + y0 + cy)
    end do
  end function
  subroutine read_records(fname, a, nrecs)
    character(*), intent(in) :: fname
    integer, intent(in) :: nrecs
    real(c_float), intent(out) :: a(nx, ny, *)
    integer :: r, u
    open(newunit=u, file=fname, access='direct', form='unformatted', recl=4*nx*ny, status='old')
    do r = 1, nrecs
       read(u, rec=r) a(:, :, r)
    end do
    close(u)
  end subroutine
end program greb_host
