! greb_c_api.f90 -- iso_c_binding interface of the MI355X GREB engine (include/greb_engine.h), shared by the hosts.
module greb_c_api
  use iso_c_binding
  implicit none

  type, bind(C) :: greb_params
     real(c_float) :: pi, sig, rho_ocean, rho_land, rho_air, cp_ocean, cp_land, cp_air, eps
     real(c_float) :: d_ocean, d_land, d_air, ct_sens, da_ice, a_no_ice, a_cloud
     real(c_float) :: Tl_ice1, Tl_ice2, To_ice1, To_ice2
     real(c_float) :: co_turb, kappa, ce, cq_latent, cq_rain, z_air, z_vapor, r_qviwv
     real(c_float) :: p_emi(10)
     real(c_float) :: co2_flux
     integer(c_int32_t) :: ipx, ipy, year0, dt, dt_crcl
  end type greb_params

  ! per-member physics of a perturbed-physics ensemble; NaN = keep the engine-wide value
  type, bind(C) :: greb_member_overrides
     real(c_float) :: da_ice, a_no_ice, a_cloud, kappa
  end type greb_member_overrides

  type, bind(C) :: greb_fields
     type(c_ptr) :: z_topo, glacier, sw_solar, tclim, qclim, uclim, vclim, mldclim, cldclim, swetclim
  end type greb_fields

  interface
     subroutine greb_params_default(p) bind(C, name="greb_params_default")
       import :: greb_params
       type(greb_params), intent(out) :: p
     end subroutine
     integer(c_int) function greb_engine_create(p, nx, ny, f, n_members, overrides, device, flags, eng) &
          bind(C, name="greb_engine_create")
       import :: greb_params, greb_fields, c_int, c_ptr
       type(greb_params), intent(in) :: p
       integer(c_int), value :: nx, ny, n_members, device, flags
       type(greb_fields), intent(in) :: f
       type(c_ptr), value :: overrides
       type(c_ptr), intent(out) :: eng
     end function
     integer(c_int) function greb_engine_flux_correction(eng, years, yearly) bind(C, name="greb_engine_flux_correction")
       import :: c_int, c_ptr, c_float
       type(c_ptr), value :: eng
       integer(c_int), value :: years
       real(c_float), intent(out) :: yearly(*)
     end function
     integer(c_int) function greb_engine_run(eng, years, co2_ppm, monthly, yearly, run_flags) bind(C, name="greb_engine_run")
       import :: c_int, c_ptr, c_float
       type(c_ptr), value :: eng
       integer(c_int), value :: years, run_flags
       real(c_float), intent(in) :: co2_ppm(*)
       real(c_float), intent(out) :: monthly(*), yearly(*)
     end function
     integer(c_int) function greb_engine_get_corrections(eng, member, corr, state5) &
          bind(C, name="greb_engine_get_corrections")
       import :: c_int, c_ptr, c_float
       type(c_ptr), value :: eng
       integer(c_int), value :: member
       real(c_float), intent(out) :: corr(*), state5(*)
     end function
     integer(c_int) function greb_engine_set_corrections(eng, member, corr, state5) &
          bind(C, name="greb_engine_set_corrections")
       import :: c_int, c_ptr, c_float
       type(c_ptr), value :: eng
       integer(c_int), value :: member
       real(c_float), intent(in) :: corr(*), state5(*)
     end function
     integer(c_int) function greb_log_exp_switches(log_exp) bind(C, name="greb_log_exp_switches")
       import :: c_int
       integer(c_int), value :: log_exp
     end function
     integer(c_int) function greb_engine_set_experiment(eng, switches) bind(C, name="greb_engine_set_experiment")
       import :: c_int, c_ptr
       type(c_ptr), value :: eng
       integer(c_int), value :: switches
     end function
     integer(c_int) function greb_engine_get_state(eng, member, state5) bind(C, name="greb_engine_get_state")
       import :: c_int, c_ptr, c_float
       type(c_ptr), value :: eng
       integer(c_int), value :: member
       real(c_float), intent(out) :: state5(*)
     end function
     integer(c_int) function greb_engine_set_state(eng, member, state5) bind(C, name="greb_engine_set_state")
       import :: c_int, c_ptr, c_float
       type(c_ptr), value :: eng
       integer(c_int), value :: member
       real(c_float), intent(in) :: state5(*)
     end function
     integer(c_int) function greb_engine_destroy(eng) bind(C, name="greb_engine_destroy")
       import :: c_int, c_ptr
       type(c_ptr), value :: eng
     end function
     function greb_engine_last_error(eng) bind(C, name="greb_engine_last_error") result(msg)
       import :: c_ptr
       type(c_ptr), value :: eng
       type(c_ptr) :: msg
     end function
  end interface
contains
  subroutine engine_check(rc, eng, what)
    integer(c_int), intent(in) :: rc
    type(c_ptr), intent(in) :: eng
    character(*), intent(in) :: what
    character(kind=c_char), pointer :: cmsg(:)
    type(c_ptr) :: p
    integer :: i
    if (rc == 0) return
    write(*, '(a,a,a,i0)') 'greb_host: ', what, ' failed, code ', rc
    p = greb_engine_last_error(eng)
    if (c_associated(p)) then
       call c_f_pointer(p, cmsg, [512])
       do i = 1, 512
          if (cmsg(i) == c_null_char) exit
          write(*, '(a)', advance='no') cmsg(i)
       end do
       write(*, *)
    end if
    error stop 1
  end subroutine
end module greb_c_api
