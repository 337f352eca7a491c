!================================================================================
! mini_commander -- a ~150-line stand-in for the Commander3 Gibbs driver that exercises the claim
! "the Fortran driver stays and calls through ISO_C_BINDING": it follows the call order of
!   sample_amps_by_CG        (commander3/src/comm_signal_mod.f90:154-216)
!     cr_computeRHS          (:176)
!     initPrecond            (:179)
!     solve_cr_eqn_by_CG     (:181)   -> update_precond (comm_cr_mod.f90:76) + PCG
!     cr_x2amp               (:182)
! for a CMB-only, one-band, temperature model (BASELINE.json configs[0]: Nside=64, lmax=128) with synthetic
! inputs.  The Fortran side owns the random draws (here: a simple LCG + Box-Muller standing in for planck_rng).
! Build: make -C fortran     Run: ./fortran/mini_commander   (needs a GPU)
!================================================================================
program mini_commander
  use iso_c_binding
  use cmdr_hip_mod
  implicit none
  integer(c_int), parameter :: nside = 64, lmax = 128, nmaps = 1
  integer, parameter        :: npix = 12*nside*nside, nalm = (lmax+1)**2
  real(c_double), parameter :: pi = 3.141592653589793238462643383279502884d0
  type(c_ptr)    :: ctx
  integer(c_int) :: ierr, band, comp, niter, stat
  real(c_double), allocatable, target :: siN(:), b_l(:), sqrtS(:), sqrtInvS(:), S(:), F_mean(:)
  real(c_double), allocatable, target :: resid(:), xi(:), eta(:), rhs(:), x(:), y(:)
  type(c_ptr)    :: resid_p(1), xi_p(1)
  real(c_double) :: res(2), sigma, Dl, Cl, z, dz, fwhm, t0, t1
  integer        :: l, i, iter
  integer(8)     :: seed

  if (cmdr_device_count() < 1) then
     write(*,*) 'mini_commander: no GPU visible (libcmdr_hip has no CPU path)'
     stop 2
  end if

  ! --- data(1): white noise rms ~ (1 + 0.5 z), Gaussian beam (gaussbeam path, comm_utils.f90:91-92)
  allocate(siN(npix), b_l(0:lmax), sqrtS(0:lmax), sqrtInvS(0:lmax), S(0:lmax), F_mean(1))
  fwhm  = 60.d0 / 60.d0 * pi / 180.d0
  sigma = fwhm / sqrt(8.d0*log(2.d0))
  do l = 0, lmax
     b_l(l) = exp(-0.5d0*l*(l+1.d0)*sigma**2)
     Dl = 1000.d0                                   ! flat D_l^TT (power_law, comm_Cl_mod.f90:226-246)
     if (l == 0) then
        Cl = Dl
     else
        Cl = Dl * 2.d0*pi / (l*(l+1.d0))            ! comm_Cl_mod.f90:332-336
     end if
     S(l) = Cl; sqrtS(l) = sqrt(Cl); sqrtInvS(l) = 1.d0/sqrt(Cl)
  end do
  do i = 1, npix                                     ! crude z per pixel is enough for a plumbing test
     z = 1.d0 - 2.d0*(i-0.5d0)/npix
     siN(i) = 1.d0 / (40.d0*(1.d0 + 0.5d0*z))
     if (abs(z) < 0.2d0) siN(i) = 0.d0               ! mask (comm_N_rms_mod.f90:179-193)
  end do
  F_mean(1) = 1.d0

  call cmdr_check(cmdr_ctx_create(0_c_int, ctx), 'cmdr_ctx_create')
  band = cmdr_band_add(ctx, nside, lmax, nmaps, siN, b_l, 1.d0, c_null_ptr, c_null_ptr)
  call cmdr_check(band, 'cmdr_band_add')
  comp = cmdr_comp_add(ctx, lmax, nmaps, lmax, c_loc(sqrtS), c_loc(sqrtInvS), c_loc(S), F_mean, 1_c_int)
  call cmdr_check(comp, 'cmdr_comp_add')
  call cmdr_check(cmdr_finalize(ctx), 'cmdr_finalize')
  if (cmdr_ncr(ctx) /= nalm) stop 'ncr mismatch'

  allocate(resid(npix), xi(npix), eta(nalm), rhs(nalm), x(nalm), y(nalm))
  seed = 163425_8                                    ! BASE_SEED, tutorial/param_tutorial.txt:15
  call cmdr_check(cmdr_precond_init_diag(ctx), 'initPrecond')        ! comm_signal_mod.f90:179

  do iter = 1, 3                                     ! Gibbs loop, commander.f90:179-254
     do i = 1, npix                                  ! compute_residual stand-in: noise-only data
        resid(i) = 0.d0
        if (siN(i) > 0.d0) resid(i) = gauss(seed) / siN(i)
        xi(i) = gauss(seed)                          ! draw order: pixel inner (comm_cr_mod.f90:602-608)
     end do
     do i = 1, nalm
        eta(i) = gauss(seed)                         ! comm_cr_mod.f90:704-709
     end do
     resid_p(1) = c_loc(resid); xi_p(1) = c_loc(xi)
     call cpu_time(t0)
     call cmdr_check(cmdr_compute_rhs(ctx, 1_c_int, resid_p, xi_p, c_loc(eta), c_null_ptr, rhs), 'cr_computeRHS')
     call cmdr_check(cmdr_precond_update_diag(ctx), 'update_precond')   ! comm_cr_mod.f90:76
     call cmdr_check(cmdr_solve(ctx, rhs, x, CMDR_CRIT_FIXED_ITER, 1.d-8, 5_c_int, 50_c_int, 1_c_int, &
          & c_null_ptr, niter, res, stat), 'solve_cr_eqn_by_CG')
     call cpu_time(t1)
     ! consistency check in Fortran: || A S^-1/2 x - b || through cr_matmulA
     y = x * reshape(spread_inv(), [nalm])
     call cmdr_check(cmdr_matmulA(ctx, y, x), 'cr_matmulA')
     write(*,'(a,i2,a,i3,a,es10.3,a,es10.3,a,es10.3,a,f7.3,a)') ' sample ', iter, ': CG iters = ', niter, &
          & '  res = ', res(1), '  delta0 = ', res(2), '  |Ax-b|/|b| = ', &
          & sqrt(sum((x-rhs)**2)/sum(rhs**2)), '  (', t1-t0, ' s)'
  end do
  call cmdr_check(cmdr_ctx_destroy(ctx), 'cmdr_ctx_destroy')
  write(*,*) 'mini_commander: OK'

contains

  function spread_inv() result(v)        ! S^-1/2 per packed a_lm index (comm_map_mod.f90:228-261 layout)
    real(c_double) :: v(nalm)
    integer :: m, ll, k
    k = 0
    do ll = 0, lmax
       k = k + 1; v(k) = sqrtInvS(ll)
    end do
    do m = 1, lmax
       do ll = m, lmax
          v(k+1) = sqrtInvS(ll); v(k+2) = sqrtInvS(ll); k = k + 2
       end do
    end do
  end function spread_inv

  function uni(s) result(u)
    integer(8), intent(inout) :: s
    real(c_double) :: u
    s = mod(s*6364136223846793005_8 + 1442695040888963407_8, 9223372036854775807_8)
    u = (real(abs(s), c_double) + 1.d0) / 9223372036854775809.d0
  end function uni

  function gauss(s) result(g)
    integer(8), intent(inout) :: s
    real(c_double) :: g, u1, u2
    u1 = uni(s); u2 = uni(s)
    g = sqrt(-2.d0*log(u1)) * cos(2.d0*pi*u2)
  end function gauss

end program mini_commander
