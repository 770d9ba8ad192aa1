! greb_host_original.f90 -- thin Fortran host for the UPSTREAM model variant's shell on the MI355X engine.
!
! Stand-in for `program time_ex` (src/greb.original.shell.web-public.f90) + the sequencing of its greb_model
! (src/greb.original.model.f90:139-233): reads ./namelist_original (groups NUMERICS: time_flux, time_ctrl,
! time_scnr; PHYSICS: log_exp) and the ten input/ files, applies the experiment's changes to the boundary data
! (:162-166), runs flux correction at CO2_ctrl, the control run (-> output/control, after the TF_correct "test
! write" of :204-206) and the scenario run (-> output/scenario) through the C ABI (host/greb_c_api.f90).
! What log_exp changes in the PROCESSES is the engine's switch set (greb_log_exp_switches, GREB_X_*).
! The yearly console line is the original's (:977): year, global mean, tsmn(48,24+3), tsmn(16,24+14).  The engine's
! yearly record carries the global mean and ONE point (its ipx/ipy = the first of the two); the second point is the
! annual mean of the same year's monthly records at (16,38) -- the records the host writes anyway: the day-weighted
! mean of the twelve monthly means is the mean over the year's 730 steps; it is free of the ~2e-4 K of rounding the
! original's own running fp32 sum of 730 values of ~300 K carries, so the two agree to that).  The
! flux-correction lines, which have no monthly records, carry the first point only.
program greb_host_original
  use iso_c_binding
  use greb_c_api
  implicit none
  integer, parameter :: nx = 96, ny = 48, nstep = 730
  integer :: time_flux, time_ctrl, time_scnr, log_exp
  namelist / numerics / time_flux, time_ctrl, time_scnr
  namelist / physics / log_exp
  type(greb_params) :: prm
  type(greb_fields) :: fld
  type(c_ptr) :: eng
  real(c_float), allocatable, target :: z_topo(:,:), glacier(:,:), sw_solar(:,:)
  real(c_float), allocatable, target :: tclim(:,:,:), qclim(:,:,:), uclim(:,:,:), vclim(:,:,:), &
       mldclim(:,:,:), cldclim(:,:,:), swetclim(:,:,:)
  real(c_float), allocatable :: monthly(:), yearly(:), yflux(:), co2(:), corr(:), start5(:), now5(:)
  integer(c_int) :: rc, x, x_noscn
  integer, parameter :: jday_mon(12) = (/31,28,31,30,31,30,31,31,30,31,30,31/)   ! src/greb.f90:42
  integer :: n, irec, nrec
  integer(8) :: off
  real :: co2_ctrl, year

  time_flux = 0; time_ctrl = 0; time_scnr = 0; log_exp = 0
  print*,'% start climate shell'
  open(10, file='namelist_original', action='read', status='old')
  read(10, nml=numerics)
  read(10, nml=physics)
  close(10)

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
  print*,'% time flux/control/scenario: ', time_flux, time_ctrl, time_scnr

  ! the original's constants: greb.f90's defaults except cp_land = cp_ocean/4.5 (greb.original.model.f90:69)
  call greb_params_default(prm)
  prm%cp_land = prm%cp_ocean / 4.5
  prm%ipx = 48; prm%ipy = 24 + 3                       ! first diagnostic point of the original's print (:977)
  co2_ctrl = 340.
  if (log_exp == 12 .or. log_exp == 13) co2_ctrl = 298. ! A1B scenario (:178-179)
  prm%co2_flux = co2_ctrl

  ! boundary data of the sensitivity experiment (:162-166)
  if (log_exp == 1) where (z_topo > 1.) z_topo = 1.0
  if (log_exp <= 2) cldclim = 0.7
  if (log_exp <= 3) qclim = 0.0052
  if (log_exp <= 9 .or. log_exp == 11) mldclim = prm%d_ocean

  fld%z_topo = c_loc(z_topo); fld%glacier = c_loc(glacier); fld%sw_solar = c_loc(sw_solar)
  fld%tclim = c_loc(tclim); fld%qclim = c_loc(qclim); fld%uclim = c_loc(uclim); fld%vclim = c_loc(vclim)
  fld%mldclim = c_loc(mldclim); fld%cldclim = c_loc(cldclim); fld%swetclim = c_loc(swetclim)
  eng = c_null_ptr
  rc = greb_engine_create(prm, nx, ny, fld, 1, c_null_ptr, 0_c_int, 0_c_int, eng)
  call engine_check(rc, eng, 'greb_engine_create')
  x = greb_log_exp_switches(int(log_exp, c_int))
  x_noscn = iand(x, not(128))                           ! GREB_X_SST_PLUS1 belongs to the scenario loop only (:224-226)
  rc = greb_engine_set_experiment(eng, x_noscn)
  call engine_check(rc, eng, 'greb_engine_set_experiment')

  print*,'% flux correction ', co2_ctrl
  allocate(yflux(2*max(time_flux,1)), start5(5*nx*ny), now5(5*nx*ny), corr(3*nstep*nx*ny))
  rc = greb_engine_flux_correction(eng, int(time_flux, c_int), yflux)
  call engine_check(rc, eng, 'greb_engine_flux_correction')
  do n = 1, time_flux
     print *, 0.0, yflux(2*n-1), yflux(2*n)
  end do
  rc = greb_engine_get_corrections(eng, 0_c_int, corr, start5) ! both runs start from here (:201 aliases Ts_ini..)
  call engine_check(rc, eng, 'greb_engine_get_corrections')

  ! "test write qflux" (:204-206), then the control run over the same unit (:208-215)
  open(21, file='output/control', access='direct', form='unformatted', recl=4*nx*ny)
  do irec = 1, nstep
     write(21, rec=irec) corr((irec-1)*nx*ny+1 : irec*nx*ny)
  end do
  print*,'% CONTROL RUN CO2=', co2_ctrl, '  time=', time_ctrl, 'yr'
  if (time_ctrl > 0) then
     nrec = time_ctrl*12*5
     allocate(monthly(int(nrec,8)*nx*ny), yearly(2*time_ctrl), co2(time_ctrl))
     co2 = co2_ctrl
     rc = greb_engine_run(eng, int(time_ctrl, c_int), co2, monthly, yearly, 0_c_int)
     call engine_check(rc, eng, 'greb_engine_run (control)')
     year = 1970.
     do n = 1, time_ctrl
        print *, year, yearly(2*n-1), yearly(2*n), second_point(n)
        year = year + 1
     end do
     do irec = 1, nrec
        off = int(irec-1, 8)*nx*ny
        write(21, rec=irec) monthly(off+1:off+nx*ny)
     end do
     deallocate(monthly, yearly, co2)
  end if
  close(21)

  ! scenario run (:217-232): state back to the end of the flux phase, cap_surf as the control run left it
  print*,'% SCENARIO EXP: ', log_exp, '  time=', time_scnr, 'yr'
  if (time_scnr > 0) then
     rc = greb_engine_get_state(eng, 0_c_int, now5)
     call engine_check(rc, eng, 'greb_engine_get_state')
     now5(1:4*nx*ny) = start5(1:4*nx*ny)
     rc = greb_engine_set_state(eng, -1_c_int, now5)
     call engine_check(rc, eng, 'greb_engine_set_state')
     rc = greb_engine_set_experiment(eng, x)
     call engine_check(rc, eng, 'greb_engine_set_experiment')
     nrec = time_scnr*12*5
     allocate(monthly(int(nrec,8)*nx*ny), yearly(2*time_scnr), co2(time_scnr))
     year = 1940.
     do n = 1, time_scnr
        co2(n) = co2_level(log_exp, year)
        if (log_exp >= 14 .and. log_exp <= 16) co2(n) = co2_ctrl   ! :225
        year = year + 1
     end do
     rc = greb_engine_run(eng, int(time_scnr, c_int), co2, monthly, yearly, 0_c_int)
     call engine_check(rc, eng, 'greb_engine_run (scenario)')
     year = 1940.
     do n = 1, time_scnr
        print *, year, yearly(2*n-1), yearly(2*n), second_point(n)
        year = year + 1
     end do
     open(22, file='output/scenario', access='direct', form='unformatted', recl=4*nx*ny)
     do irec = 1, nrec
        off = int(irec-1, 8)*nx*ny
        write(22, rec=irec) monthly(off+1:off+nx*ny)
     end do
     close(22)
  end if
  rc = greb_engine_destroy(eng)

contains
  ! tsmn(16,24+14) - 273.15 of year yr (1-based) of the run whose monthly records are in `monthly`
  real function second_point(yr)
    integer, intent(in) :: yr
    integer :: mm
    integer(8) :: o
    double precision :: acc
    acc = 0.d0
    do mm = 1, 12
       o = (int(yr-1, 8)*12 + (mm-1))*5*nx*ny + int(24+14-1, 8)*nx + 16   ! record Tsurf of that month, point (16,38)
       acc = acc + dble(2*jday_mon(mm))*dble(monthly(o))
    end do
    second_point = real(acc/dble(nstep)) - 273.15
  end function
  ! co2_level of the original (:939-951)
  real function co2_level(le, yr)
    integer, intent(in) :: le
    real, intent(in) :: yr
    real :: co2_1950, co2_2000, co2_2050
    co2_level = 680.
    if (le == 12 .or. le == 13) then
       co2_1950 = 310.;  co2_2000 = 370.;  co2_2050 = 520.
       if (yr <= 2000.) co2_level = co2_1950 + 60./50.*(yr-1950.)
       if (yr > 2000. .and. yr <= 2050.) co2_level = co2_2000 + 150./50.*(yr-2000.)
       if (yr > 2050. .and. yr <= 2100.) co2_level = co2_2050 + 180./50.*(yr-2050.)
    end if
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
end program greb_host_original
