!================================================================================
! api_tour -- calls, from Fortran through ISO_C_BINDING, the parts of the C ABI that mini_commander does not touch:
! two bands, two diffuse components (one with a spatially varying mixing matrix), a template block and a point-source
! block (compact components), the error path (a preconditioner type that rejects compact components), getSigmaL, the
! chain-file a_lm order, and the C_l step with the update entry points (sampleCls, updateS, set_cl, set_active).  Tiny sizes (Nside 8, lmax 16); every check is a size-independent property.
! Build: make -C fortran     Run: ./fortran/api_tour   (needs a GPU)
!================================================================================
program api_tour
  use iso_c_binding
  use cmdr_hip_mod
  implicit none
  integer(c_int), parameter :: nside = 8, lmax = 16, nmaps = 1, nband = 2
  integer, parameter        :: npix = 12*nside*nside, nalm = (lmax+1)**2
  real(c_double), parameter :: pi = 3.141592653589793238462643383279502884d0
  type(c_ptr)    :: ctx
  integer(c_int) :: ierr, ib, ic, blk, niter, stat
  integer(c_int64_t) :: ncr
  real(c_double), allocatable, target :: siN(:), b_l(:), sqrtS(:), sqrtInvS(:), S(:), F(:), Fmap(:), z(:)
  real(c_double), allocatable, target :: x(:), y(:), ax(:), ay(:), sol(:), sig(:)
  real(c_float),  allocatable         :: chain(:)
  real(c_double), allocatable         :: back(:)
  integer(c_int64_t), allocatable     :: cell(:)
  integer(c_int),     allocatable     :: par(:)
  real(c_double),     allocatable     :: val(:)
  real(c_double) :: res(2), sgm(2), mean(2), lhs, rhs
  real(c_double), allocatable, target :: Dl(:)
  real(c_double), allocatable :: u(:)
  type(cmdr_cl_bin), allocatable :: bins(:)
  integer(c_int) :: nused, niter2
  integer(c_size_t) :: mem_free, mem_total, nb
  type(c_ptr) :: dx, dy, dz
  integer :: l, i, n
  integer(8) :: seed
  character(kind=c_char) :: rid(128)
  real(c_double), allocatable, target :: siN_low(:)
  integer(c_int) :: nside_low(2)
  type(c_ptr)    :: low_p(2)
  real(c_double) :: pms(6)
  integer(c_long_long) :: pcnt(6)

  if (cmdr_device_count() < 1) then
     write(*,*) 'api_tour: no GPU visible (libcmdr_hip has no CPU path)'
     stop 2
  end if
  allocate(siN(npix), z(npix), b_l(0:lmax), sqrtS(0:lmax), sqrtInvS(0:lmax), S(0:lmax), F(nband), Fmap(npix))
  do i = 1, npix
     z(i) = 1.d0 - 2.d0*(i-0.5d0)/npix
     siN(i) = 1.d0 / (3.d0*(1.d0 + 0.5d0*z(i)))
  end do
  do l = 0, lmax
     b_l(l) = exp(-0.5d0*l*(l+1.d0)*(0.05d0)**2)
     S(l) = 100.d0 / max(l*(l+1.d0), 1.d0); sqrtS(l) = sqrt(S(l)); sqrtInvS(l) = 1.d0/sqrtS(l)
  end do

  call cmdr_check(cmdr_set_device(0_c_int), 'cmdr_set_device')        ! one rank per GPU: the node-local rank goes here
  call cmdr_check(cmdr_ctx_create(0_c_int, ctx), 'cmdr_ctx_create')
  do ib = 1, nband
     call cmdr_check(cmdr_band_add(ctx, nside, lmax, nmaps, siN, b_l, 1.d0, c_null_ptr, c_null_ptr), 'cmdr_band_add')
  end do
  F = 1.d0                                                           ! component 0: CMB-like, constant mixing
  call cmdr_check(cmdr_comp_add(ctx, lmax, nmaps, lmax, c_loc(sqrtS), c_loc(sqrtInvS), c_loc(S), F, 1_c_int), 'comp 0')
  ! compact block between the two diffuse components (compList order = stacked-vector order): 2 templates on band 0
  sgm = [20.d0, 5.d0]; mean = [1.d0, -1.d0]
  blk = cmdr_compact_add(ctx, 2_c_int, sgm, mean, 1_c_int)
  call cmdr_check(blk, 'cmdr_compact_add')
  n = 2*npix
  allocate(cell(n), par(n), val(n))
  do i = 1, npix
     cell(i) = i-1;        par(i) = 0;        val(i) = 1.d0           ! monopole
     cell(npix+i) = i-1;   par(npix+i) = 1;   val(npix+i) = z(i)      ! dipole-z
  end do
  call cmdr_check(cmdr_compact_set_band(ctx, blk, 0_c_int, int(n, c_int64_t), cell, par, val), 'compact band 0')
  F = [1.d0, 0.3d0]                                                  ! component 1: varying mixing on both bands
  ic = cmdr_comp_add(ctx, lmax, nmaps, lmax, c_loc(sqrtS), c_loc(sqrtInvS), c_loc(S), F, 1_c_int)
  call cmdr_check(ic, 'comp 1')
  do ib = 1, nband
     Fmap = F(ib) * (1.d0 + 0.1d0*z)
     call cmdr_check(cmdr_comp_set_mixing_map(ctx, ic, int(ib-1, c_int), c_loc(Fmap), nmaps), 'mixing map')
  end do
  ! one point source seen by both bands: a 5-pixel footprint
  blk = cmdr_compact_add(ctx, 1_c_int, [3.d0], [0.d0], 1_c_int)
  do ib = 1, nband
     call cmdr_check(cmdr_compact_set_band(ctx, blk, int(ib-1, c_int), 5_c_int64_t, &
          & int([100, 101, 102, 131, 132], c_int64_t), int([0, 0, 0, 0, 0], c_int), &
          & [0.2d0, 1.d0, 0.2d0, 0.5d0, 0.5d0]*ib), 'source footprint')
  end do
  call cmdr_check(cmdr_finalize(ctx), 'cmdr_finalize')
  ncr = cmdr_ncr(ctx)
  if (ncr /= 2*nalm + 3) stop 'api_tour: ncr mismatch'

  ! A is symmetric (unit ring weights) and >= 1
  allocate(x(ncr), y(ncr), ax(ncr), ay(ncr), sol(ncr))
  seed = 4242_8
  do i = 1, int(ncr)
     x(i) = uni(seed) - 0.5d0; y(i) = uni(seed) - 0.5d0
  end do
  call cmdr_check(cmdr_matmulA(ctx, x, ax), 'cr_matmulA x')
  call cmdr_check(cmdr_matmulA(ctx, y, ay), 'cr_matmulA y')
  lhs = sum(y*ax); rhs = sum(x*ay)
  if (abs(lhs-rhs) > 1.d-10*sqrt(sum(y*y)*sum(ax*ax))) stop 'api_tour: A not symmetric'
  if (sum(x*ax) < sum(x*x)*(1.d0-1.d-12)) stop 'api_tour: A < 1'

  ! diagonal preconditioner + compact dense blocks; solve A sol = A x  =>  S^1/2-scaled x comes back after enough iterations
  call cmdr_check(cmdr_precond_init_diag(ctx), 'initPrecond')
  call cmdr_check(cmdr_precond_update_diag(ctx), 'update_precond')
  call cmdr_check(cmdr_solve(ctx, ax, sol, CMDR_CRIT_RESIDUAL, 1.d-12, 5_c_int, 400_c_int, 1_c_int, c_null_ptr, &
       & niter, res, stat), 'solve_cr_eqn_by_CG')
  if (stat /= 0) stop 'api_tour: CG did not converge'
  write(*,'(a,i4,a,es10.3)') ' converged in ', niter, ' iterations, delta/delta0 = ', res(1)/res(2)

  ! the same through device-resident vectors: x and A x stay in HBM between calls (what a chain does inside one solve)
  call cmdr_check(cmdr_dev_mem_info(mem_free, mem_total), 'mem_info')
  if (mem_free <= 0 .or. mem_free > mem_total) stop 'api_tour: mem_info'
  nb = int(ncr, c_size_t) * 8_c_size_t
  call cmdr_check(cmdr_dev_alloc(nb, dx), 'dev_alloc x'); call cmdr_check(cmdr_dev_alloc(nb, dy), 'dev_alloc y')
  call cmdr_check(cmdr_dev_alloc(nb, dz), 'dev_alloc z')
  call cmdr_check(cmdr_host_register(c_loc(x), nb), 'host_register')       ! long-lived host array: copies by DMA from now on
  call cmdr_check(cmdr_memcpy_h2d(dx, c_loc(x), nb), 'h2d')
  call cmdr_check(cmdr_host_unregister(c_loc(x)), 'host_unregister')
  call cmdr_check(cmdr_matmulA_dev(ctx, dx, dy), 'cr_matmulA_dev')
  call cmdr_check(cmdr_memcpy_d2h(c_loc(ay), dy, nb), 'd2h')
  if (maxval(abs(ay - ax)) > 0.d0) stop 'api_tour: device-pointer matvec differs from the host-pointer one'
  call cmdr_check(cmdr_solve_dev(ctx, dy, dz, CMDR_CRIT_RESIDUAL, 1.d-12, 5_c_int, 400_c_int, 1_c_int, c_null_ptr, &
       & niter2, res, stat), 'solve_dev')
  call cmdr_check(cmdr_memcpy_d2h(c_loc(ay), dz, nb), 'd2h solution')
  if (stat /= 0 .or. niter2 /= niter) stop 'api_tour: solve_dev'
  if (maxval(abs(ay - sol)) > 1.d-12*maxval(abs(sol))) stop 'api_tour: solve_dev differs from solve'
  call cmdr_check(cmdr_device_synchronize(), 'sync')
  call cmdr_check(cmdr_dev_free(dx), 'free'); call cmdr_check(cmdr_dev_free(dy), 'free'); call cmdr_check(cmdr_dev_free(dz), 'free')
  if (cmdr_band_npix(ctx, 0_c_int) /= npix) stop 'api_tour: band_npix'

  ! error path: the pseudo-inverse preconditioner rejects compact components; the message comes through cmdr_last_error
  ierr = cmdr_precond_init_pseudoinv(ctx)
  if (ierr >= 0) stop 'api_tour: expected an error'

  ! getSigmaL and the chain-file order
  allocate(sig(0:lmax), chain(nalm), back(nalm))
  call cmdr_check(cmdr_sigma_l(sol(1:nalm), lmax, nmaps, sig), 'getSigmaL')
  if (any(sig < 0.d0)) stop 'api_tour: sigma_l < 0'
  call cmdr_check(cmdr_alm_to_chain_order(sol(1:nalm), lmax, nmaps, chain), 'chain order')
  call cmdr_check(cmdr_alm_from_chain_order(chain, lmax, nmaps, back), 'chain order back')
  if (maxval(abs(back - real(real(sol(1:nalm), c_float), c_double))) > 0.d0) stop 'api_tour: chain order round trip'
  if (abs(chain(1) - real(sol(1), c_float)) > 0.0) stop 'api_tour: chain order l=0'      ! (l,m)=(0,0) is index 0 in both

  ! C_l | a_lm for component 0 and the next amplitude sample with the new prior (commander.f90:229 sample_powspec)
  allocate(Dl(0:lmax), bins(3), u(3))
  do l = 0, lmax
     Dl(l) = S(l) * l*(l+1) / (2.d0*pi)
  end do
  Dl(0) = S(0)
  bins(1) = cmdr_cl_bin(2, 5, 1, 1, 0.1d0*Dl(2)); bins(2) = cmdr_cl_bin(6, 11, 1, 1, 0.1d0*Dl(6))
  bins(3) = cmdr_cl_bin(12, lmax, 1, 0, 0.d0)                        ! not sampled
  u = [uni(seed), uni(seed), uni(seed)]
  lhs = Dl(12)
  ierr = cmdr_cl_sample_binned(lmax, nmaps, sig, S, [1.d0], 3_c_int, bins, u, 3_c_int, Dl, nused)
  call cmdr_check(ierr, 'sampleCls')
  if (ierr /= 0 .or. nused /= 2) stop 'api_tour: sampleCls failed'
  if (Dl(12) /= lhs .or. Dl(2) <= 0.d0 .or. Dl(3) /= Dl(2) .or. Dl(6) == Dl(2)) stop 'api_tour: sampled D_l not as binned'
  ierr = cmdr_cl_update_S(lmax, nmaps, 0_c_int, Dl, [1.d0], sqrtS, sqrtInvS, S)
  call cmdr_check(ierr, 'updateS')
  if (ierr /= 0) stop 'api_tour: updateS found a non-positive-definite multipole'
  if (abs(sqrtS(2)**2 - S(2)) > 1.d-14*S(2)) stop 'api_tour: sqrtS**2 /= S'
  call cmdr_check(cmdr_comp_set_cl(ctx, 0_c_int, sqrtS, sqrtInvS, S), 'set_cl')
  call cmdr_check(cmdr_compact_set_active(ctx, blk, 0_c_int), 'compact inactive')      ! next sampling group
  call cmdr_check(cmdr_precond_init_diag(ctx), 'initPrecond 2')
  call cmdr_check(cmdr_precond_update_diag(ctx), 'update_precond 2')
  call cmdr_check(cmdr_matmulA(ctx, x, ax), 'cr_matmulA, new prior')
  if (ax(ncr) /= 0.d0) stop 'api_tour: inactive block not zero'
  call cmdr_check(cmdr_solve(ctx, ax, sol, CMDR_CRIT_RESIDUAL, 1.d-12, 5_c_int, 400_c_int, 1_c_int, c_null_ptr, &
       & niter, res, stat), 'second solve')
  if (stat /= 0) stop 'api_tour: second CG did not converge'
  write(*,'(a,i4,a)') ' second sample (new C_l) converged in ', niter, ' iterations'

  ! ---- round-2 entry points ------------------------------------------------------------------------------------
  ! RCCL inside the library with a communicator of one rank: the id would be MPI_Bcast by the driver (INTEGRATION.md)
  if (cmdr_rccl_version() > 0) then
     call cmdr_check(cmdr_rccl_unique_id(rid), 'rccl id')
     call cmdr_check(cmdr_ctx_init_rccl(ctx, rid, 0_c_int, 1_c_int), 'rccl init')
     if (cmdr_ctx_rccl_size(ctx) /= 1) stop 'api_tour: rccl communicator size'
     call cmdr_check(cmdr_matmulA(ctx, x, ay), 'cr_matmulA through ncclAllReduce')
     if (any(ay /= ax)) stop 'api_tour: a one-rank all-reduce changed the matvec'
  else
     write(*,*) 'api_tour: librccl not loadable, RCCL leg skipped'
  end if
  ! literal pmap%alm re-use of cr_matmulA (both components have the same lmax here: no effect, by construction)
  call cmdr_check(cmdr_ctx_set_literal_quirks(ctx, 1_c_int), 'literal quirks on')
  call cmdr_check(cmdr_matmulA(ctx, x, ay), 'cr_matmulA, literal')
  if (any(ay /= ax)) stop 'api_tour: literal_quirks changed A although all lmax are equal'
  call cmdr_check(cmdr_ctx_set_literal_quirks(ctx, 0_c_int), 'literal quirks off')
  ! low-l dense preconditioner block (CG_LMAX_PRECOND = 3): coadded noise at Nside 4 = sqrt of the summed siN^2 of the children
  allocate(siN_low(12*4*4))
  siN_low = sqrt(4.d0) * sum(siN) / npix                             ! smooth noise: every coarse pixel alike
  nside_low = 4
  low_p = [c_loc(siN_low), c_loc(siN_low)]
  call cmdr_check(cmdr_precond_set_lowl(ctx, 0_c_int, 3_c_int, nside_low, low_p), 'set_lowl')
  call cmdr_check(cmdr_precond_update_diag(ctx), 'update_precond with the low-l block')
  call cmdr_check(cmdr_precond_set_lowl(ctx, 0_c_int, -1_c_int, nside_low, low_p), 'lowl off (first)')
  call cmdr_check(cmdr_invM(ctx, x, ax), 'cr_invM, diagonal only')
  call cmdr_check(cmdr_precond_set_lowl(ctx, 0_c_int, 3_c_int, nside_low, low_p), 'set_lowl again')
  call cmdr_check(cmdr_precond_update_diag(ctx), 'update_precond with the low-l block')
  call cmdr_check(cmdr_invM(ctx, x, ay), 'cr_invM with the low-l block')
  ! applyLowlPrecond overwrites exactly the (L+1)^2 = 16 temperature entries with l <= 3 of component 0 (the block acts
  ! on the INPUT vector, comm_cr_mod.f90:1064-1067; with a second diffuse component M^-1 is then no longer symmetric --
  ! the reference enables the block for CMB-only sampling groups)
  if (count(ay /= ax) /= 16) stop 'api_tour: the low-l block must change 16 entries'
  if (any(ay /= ay)) stop 'api_tour: NaN from the low-l block'
  call cmdr_check(cmdr_precond_set_lowl(ctx, 0_c_int, -1_c_int, nside_low, low_p), 'lowl off')
  ! chain file: write component 0 of the sample, read it back (single precision on disk)
  ierr = cmdr_chain_write_comp('api_tour_chain.h5'//c_null_char, 1_c_int, 'cmb'//c_null_char, sol(1:nalm), lmax, nmaps, &
       & c_null_ptr, c_loc(sig), c_loc(Dl))
  if (ierr == 0) then
     call cmdr_check(cmdr_chain_read_comp('api_tour_chain.h5'//c_null_char, 1_c_int, 'cmb'//c_null_char, lmax, nmaps, &
          & c_null_ptr, back, c_null_ptr), 'chain read')
     if (maxval(abs(back - real(real(sol(1:nalm), c_float), c_double))) > 0.d0) stop 'api_tour: chain file round trip'
  else
     write(*,*) 'api_tour: libhdf5 not loadable, chain-file leg skipped'
  end if
  ! per-kernel timings incl. the matrix-unit adjoint (kinds 4, 5)
  call cmdr_check(cmdr_profile_enable(ctx, 1_c_int), 'profile on')
  call cmdr_check(cmdr_matmulA(ctx, x, ay), 'cr_matmulA, profiled')
  call cmdr_check(cmdr_profile_read_ext(ctx, 6_c_int, pms, pcnt), 'profile read')
  if (pcnt(4) /= 1 .or. pcnt(5) + pcnt(6) < 1) stop 'api_tour: profile counts'

  call cmdr_check(cmdr_ctx_destroy(ctx), 'cmdr_ctx_destroy')
  write(*,*) 'api_tour: OK'

contains

  function uni(s) result(u)
    integer(8), intent(inout) :: s
    real(c_double) :: u
    s = mod(s*6364136223846793005_8 + 1442695040888963407_8, 9223372036854775807_8)
    u = (real(abs(s), c_double) + 1.d0) / 9223372036854775809.d0
  end function uni

end program api_tour
