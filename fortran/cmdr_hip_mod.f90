!================================================================================
! cmdr_hip_mod -- ISO_C_BINDING interface to libcmdr_hip.so (include/cmdr_hip.h),
! written in the style of Commander3's own libsharp2 shim (commander3/src/sharp.f90:32-105).
!
! A Commander3 maintainer adds this file to commander3/src/, links -lcmdr_hip, and replaces the bodies of
!   cr_matmulA / cr_invM / cr_computeRHS / solve_cr_eqn_by_CG (commander3/src/comm_cr_mod.f90)
! by calls to the cmdr_* routines below (see INTEGRATION.md for the exact call sites).  The Fortran driver keeps
! ownership of the RNG (planck_rng), the component list and the parameter file; only arrays cross the boundary.
!================================================================================
module cmdr_hip_mod
  use iso_c_binding
  implicit none

  integer(c_int), parameter :: CMDR_YtW = 0, CMDR_Y = 1, CMDR_Yt = 2, CMDR_WY = 3   ! sharp.f90:8-14
  integer(c_int), parameter :: CMDR_CRIT_RESIDUAL = 0, CMDR_CRIT_FIXED_ITER = 1, CMDR_CRIT_CHISQ = 2   ! cpar%cg_conv_crit
  integer(c_int), parameter :: CMDR_MONO_PRIOR_MONOPOLE = 1, CMDR_MONO_PRIOR_MONOPOLE_DIPOLE = 2       ! mono_prior_type

  ! one node of comm_Cl's bins2 tree (comm_Cl_mod.f90:41-47), flattened depth-first; sample /= 0 where stat == 'S'
  type, bind(c) :: cmdr_cl_bin
     integer(c_int) :: lmin, lmax, spec, sample
     real(c_double) :: sigma
  end type cmdr_cl_bin

  interface
     function cmdr_last_error() bind(c, name='cmdr_last_error') result(msg)
       import :: c_ptr
       type(c_ptr) :: msg
     end function cmdr_last_error

     function cmdr_device_count() bind(c, name='cmdr_device_count') result(n)
       import :: c_int
       integer(c_int) :: n
     end function cmdr_device_count

     ! ---- SHT level (replaces sharp_make_*_info + sharp_execute, sharp.f90:44-104)
     function cmdr_sht_plan_create(nside, lmax, nrings, rings, wring, max_maps, plan) &
          & bind(c, name='cmdr_sht_plan_create') result(ierr)
       import :: c_int, c_ptr, c_double
       integer(c_int), value        :: nside, lmax, nrings, max_maps
       type(c_ptr),    value        :: rings     ! c_loc(info%rings) or c_null_ptr
       type(c_ptr),    value        :: wring     ! c_loc(info%W(:,1)) or c_null_ptr
       type(c_ptr),    intent(out)  :: plan
       integer(c_int)               :: ierr
     end function cmdr_sht_plan_create

     function cmdr_sht_plan_destroy(plan) bind(c, name='cmdr_sht_plan_destroy') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: plan
       integer(c_int)     :: ierr
     end function cmdr_sht_plan_destroy

     function cmdr_sht_execute(plan, job, nmaps, alm, map) bind(c, name='cmdr_sht_execute') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr),    value      :: plan
       integer(c_int), value      :: job, nmaps
       type(c_ptr),    intent(in) :: alm(*), map(*)     ! one column pointer per map, as sharp.f90:203-224
       integer(c_int)             :: ierr
     end function cmdr_sht_execute

     ! ---- CR level
     function cmdr_ctx_create(device, ctx) bind(c, name='cmdr_ctx_create') result(ierr)
       import :: c_int, c_ptr
       integer(c_int), value       :: device
       type(c_ptr),    intent(out) :: ctx
       integer(c_int)              :: ierr
     end function cmdr_ctx_create

     function cmdr_ctx_destroy(ctx) bind(c, name='cmdr_ctx_destroy') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int)     :: ierr
     end function cmdr_ctx_destroy

     function cmdr_ctx_set_rings(ctx, nside, nrings, rings) bind(c, name='cmdr_ctx_set_rings') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr),    value      :: ctx
       integer(c_int), value      :: nside, nrings
       integer(c_int), intent(in) :: rings(nrings)
       integer(c_int)             :: ierr
     end function cmdr_ctx_set_rings

     function cmdr_ctx_set_allreduce(ctx, fn, user) bind(c, name='cmdr_ctx_set_allreduce') result(ierr)
       import :: c_int, c_ptr, c_funptr
       type(c_ptr),    value :: ctx, user
       type(c_funptr), value :: fn          ! subroutine(user, dev_ptr, n) bind(c)
       integer(c_int)        :: ierr
     end function cmdr_ctx_set_allreduce

     ! stream-ordered variant (RCCL-style): fn(user, dev_ptr, n, hip_stream) enqueues the sum on hip_stream
     function cmdr_ctx_set_allreduce_stream(ctx, fn, user) bind(c, name='cmdr_ctx_set_allreduce_stream') result(ierr)
       import :: c_int, c_ptr, c_funptr
       type(c_ptr),    value :: ctx, user
       type(c_funptr), value :: fn
       integer(c_int)        :: ierr
     end function cmdr_ctx_set_allreduce_stream

     ! band x ring-set hybrid sharding: rings_fn sums over the ranks holding the same bands
     function cmdr_ctx_set_band_sharding(ctx, rings_fn, user, ring_replicas) &
          & bind(c, name='cmdr_ctx_set_band_sharding') result(ierr)
       import :: c_int, c_ptr, c_funptr
       type(c_ptr),    value :: ctx, user
       type(c_funptr), value :: rings_fn
       integer(c_int), value :: ring_replicas
       integer(c_int)        :: ierr
     end function cmdr_ctx_set_band_sharding

     ! RCCL inside the library: id = 128 bytes from cmdr_rccl_unique_id on ONE rank, MPI_Bcast by the driver, then
     ! every rank calls cmdr_ctx_init_rccl (collective).  All sums over ranks of the CR path then run as ncclAllReduce
     ! on the library's own stream: what mpi_dot_product / libsharp2's exchange do in comm_cr_mod (comm_utils.f90:599-614)
     function cmdr_rccl_unique_id(out128) bind(c, name='cmdr_rccl_unique_id') result(ierr)
       import :: c_int, c_char
       character(kind=c_char), intent(out) :: out128(128)
       integer(c_int)                      :: ierr
     end function cmdr_rccl_unique_id

     function cmdr_rccl_version() bind(c, name='cmdr_rccl_version') result(v)
       import :: c_int
       integer(c_int) :: v
     end function cmdr_rccl_version

     function cmdr_ctx_init_rccl(ctx, id128, rank, nranks) bind(c, name='cmdr_ctx_init_rccl') result(ierr)
       import :: c_int, c_ptr, c_char
       type(c_ptr),    value              :: ctx
       character(kind=c_char), intent(in) :: id128(128)
       integer(c_int), value              :: rank, nranks
       integer(c_int)                     :: ierr
     end function cmdr_ctx_init_rccl

     ! band x ring-set hybrid without callbacks: ncclCommSplit(color = band_group, key = ring_index); collective
     function cmdr_ctx_rccl_split_rings(ctx, band_group, ring_index, ring_replicas) &
          & bind(c, name='cmdr_ctx_rccl_split_rings') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr),    value :: ctx
       integer(c_int), value :: band_group, ring_index, ring_replicas
       integer(c_int)        :: ierr
     end function cmdr_ctx_rccl_split_rings

     ! m-sliced CG vectors inside cmdr_solve (rank of nranks keeps its range of x, r, d, q, s; reduce-scatter + all-gather)
     function cmdr_ctx_set_vector_slicing(ctx, rank, nranks) bind(c, name='cmdr_ctx_set_vector_slicing') result(ierr)
       import :: c_ptr, c_int
       type(c_ptr),    value :: ctx
       integer(c_int), value :: rank, nranks
       integer(c_int)        :: ierr
     end function cmdr_ctx_set_vector_slicing

     function cmdr_ctx_drop_rccl(ctx) bind(c, name='cmdr_ctx_drop_rccl') result(ierr)
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
       integer(c_int) :: ierr
     end function cmdr_ctx_drop_rccl

     function cmdr_ctx_rccl_size(ctx) bind(c, name='cmdr_ctx_rccl_size') result(n)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int)     :: n
     end function cmdr_ctx_rccl_size

     ! chain-file I/O of one component's sample (/<iter>/<label>/amp_alm, amp_lmax, amp_nmaps, sigma_l, Dl)
     function cmdr_chain_write_comp(chainfile, iter, label, alm, lmax, nmaps, unit_scale, sigma_l, Dl) &
          & bind(c, name='cmdr_chain_write_comp') result(ierr)
       import :: c_int, c_ptr, c_double, c_char
       character(kind=c_char), intent(in) :: chainfile(*), label(*)      ! null-terminated
       integer(c_int), value              :: iter, lmax, nmaps
       real(c_double), intent(in)         :: alm(*)
       type(c_ptr),    value              :: unit_scale, sigma_l, Dl     ! c_loc(...) or c_null_ptr
       integer(c_int)                     :: ierr
     end function cmdr_chain_write_comp

     function cmdr_chain_read_comp(chainfile, iter, label, lmax, nmaps, unit_scale, alm, Dl) &
          & bind(c, name='cmdr_chain_read_comp') result(ierr)
       import :: c_int, c_ptr, c_double, c_char
       character(kind=c_char), intent(in) :: chainfile(*), label(*)
       integer(c_int), value              :: iter, lmax, nmaps
       type(c_ptr),    value              :: unit_scale, Dl
       real(c_double), intent(out)        :: alm(*)
       integer(c_int)                     :: ierr
     end function cmdr_chain_read_comp

     ! include/cmdr_sharp.h: the sum over one communicator that sharp_execute_mpi_fortran needs when a chain has more
     ! than one rank (fn(user, host_buf, n) = MPI_Allreduce(MPI_IN_PLACE, host_buf, n, MPI_DOUBLE_PRECISION, MPI_SUM, comm))
     subroutine cmdr_sharp_register_comm(comm, fn, user) bind(c, name='cmdr_sharp_register_comm')
       import :: c_int, c_ptr, c_funptr
       integer(c_int), value :: comm
       type(c_funptr), value :: fn
       type(c_ptr),    value :: user
     end subroutine cmdr_sharp_register_comm

     function cmdr_profile_read_ext(ctx, nkinds, ms_sum, count) bind(c, name='cmdr_profile_read_ext') result(ierr)
       import :: c_int, c_ptr, c_double, c_long_long
       type(c_ptr),    value         :: ctx
       integer(c_int), value         :: nkinds
       real(c_double), intent(out)   :: ms_sum(*)
       integer(c_long_long), intent(out) :: count(*)
       integer(c_int)                :: ierr
     end function cmdr_profile_read_ext

     function cmdr_ctx_set_literal_quirks(ctx, on) bind(c, name='cmdr_ctx_set_literal_quirks') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr),    value :: ctx
       integer(c_int), value :: on
       integer(c_int)        :: ierr
     end function cmdr_ctx_set_literal_quirks

     function cmdr_band_add(ctx, nside, lmax, nmaps, siN, b_l, mb_eff, sg_mask, wring) &
          & bind(c, name='cmdr_band_add') result(idx)
       import :: c_int, c_ptr, c_double
       type(c_ptr),    value      :: ctx
       integer(c_int), value      :: nside, lmax, nmaps
       real(c_double), intent(in) :: siN(*), b_l(*)
       real(c_double), value      :: mb_eff
       type(c_ptr),    value      :: sg_mask, wring      ! c_loc(...) or c_null_ptr
       integer(c_int)             :: idx
     end function cmdr_band_add

     function cmdr_comp_add(ctx, lmax_amp, nmaps, lmax_cl, sqrtS_mat, sqrtInvS_mat, S_mat, F_mean, active) &
          & bind(c, name='cmdr_comp_add') result(idx)
       import :: c_int, c_ptr, c_double
       type(c_ptr),    value      :: ctx
       integer(c_int), value      :: lmax_amp, nmaps, lmax_cl, active
       type(c_ptr),    value      :: sqrtS_mat, sqrtInvS_mat, S_mat   ! c_loc(c%Cl%sqrtS_mat) ... or c_null_ptr
       real(c_double), intent(in) :: F_mean(*)
       integer(c_int)             :: idx
     end function cmdr_comp_add

     function cmdr_finalize(ctx) bind(c, name='cmdr_finalize') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int)     :: ierr
     end function cmdr_finalize

     function cmdr_ncr(ctx) bind(c, name='cmdr_ncr') result(n)
       import :: c_int64_t, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int64_t) :: n
     end function cmdr_ncr

     function cmdr_precond_init_diag(ctx) bind(c, name='cmdr_precond_init_diag') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int)     :: ierr
     end function cmdr_precond_init_diag

     function cmdr_precond_update_diag(ctx) bind(c, name='cmdr_precond_update_diag') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int)     :: ierr
     end function cmdr_precond_update_diag

     ! CG_LMAX_PRECOND: low-l dense preconditioner block (updateLowlPrecond / applyLowlPrecond); siN_lowres = array of
     ! c_loc(data(b)%N%siN_lowres%map), nside_lowres(b) = data(b)%N%nside_chisq_lowres
     function cmdr_precond_set_lowl(ctx, comp, lmax_pre_lowl, nside_lowres, siN_lowres) &
          & bind(c, name='cmdr_precond_set_lowl') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr),    value      :: ctx
       integer(c_int), value      :: comp, lmax_pre_lowl
       integer(c_int), intent(in) :: nside_lowres(*)
       type(c_ptr),    intent(in) :: siN_lowres(*)
       integer(c_int)             :: ierr
     end function cmdr_precond_set_lowl

     function cmdr_precond_init_pseudoinv(ctx) bind(c, name='cmdr_precond_init_pseudoinv') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int)     :: ierr
     end function cmdr_precond_init_pseudoinv

     function cmdr_precond_update_pseudoinv(ctx) bind(c, name='cmdr_precond_update_pseudoinv') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int)     :: ierr
     end function cmdr_precond_update_pseudoinv

     ! F(band,0)%p%map of a component with spatially varying mixing; F = c_null_ptr returns to the F_mean path
     function cmdr_comp_set_mixing_map(ctx, comp, band, F, nmaps) bind(c, name='cmdr_comp_set_mixing_map') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr),    value :: ctx, F
       integer(c_int), value :: comp, band, nmaps
       integer(c_int)        :: ierr
     end function cmdr_comp_set_mixing_map

     function cmdr_comp_set_cl_diag(ctx, comp, cl) bind(c, name='cmdr_comp_set_cl_diag') result(ierr)
       import :: c_int, c_ptr, c_double
       type(c_ptr),    value      :: ctx
       integer(c_int), value      :: comp
       real(c_double), intent(in) :: cl(*)
       integer(c_int)             :: ierr
     end function cmdr_comp_set_cl_diag

     ! ---- between Gibbs iterations (after cmdr_finalize)
     ! sampleCls -> updateS (comm_Cl_mod.f90:838-863): new S tables of a component
     function cmdr_comp_set_cl(ctx, comp, sqrtS_mat, sqrtInvS_mat, S_mat) bind(c, name='cmdr_comp_set_cl') result(ierr)
       import :: c_int, c_ptr, c_double
       type(c_ptr),    value      :: ctx
       integer(c_int), value      :: comp
       real(c_double), intent(in) :: sqrtS_mat(*), sqrtInvS_mat(*), S_mat(*)
       integer(c_int)             :: ierr
     end function cmdr_comp_set_cl

     function cmdr_comp_set_f_mean(ctx, comp, F_mean) bind(c, name='cmdr_comp_set_f_mean') result(ierr)
       import :: c_int, c_ptr, c_double
       type(c_ptr),    value      :: ctx
       integer(c_int), value      :: comp
       real(c_double), intent(in) :: F_mean(*)
       integer(c_int)             :: ierr
     end function cmdr_comp_set_f_mean

     ! c%active_samp_group(samp_group)
     function cmdr_comp_set_active(ctx, comp, active) bind(c, name='cmdr_comp_set_active') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr),    value :: ctx
       integer(c_int), value :: comp, active
       integer(c_int)        :: ierr
     end function cmdr_comp_set_active

     function cmdr_compact_set_active(ctx, block, active) bind(c, name='cmdr_compact_set_active') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr),    value :: ctx
       integer(c_int), value :: block, active
       integer(c_int)        :: ierr
     end function cmdr_compact_set_active

     ! comm_Cl%updateS (comm_Cl_mod.f90:316-384); returns the number of non-positive-definite multipoles
     function cmdr_cl_update_S(lmax, nmaps, lmin, Dl, RJ2unit, sqrtS_mat, sqrtInvS_mat, S_mat) &
          & bind(c, name='cmdr_cl_update_S') result(ierr)
       import :: c_int, c_double
       integer(c_int), value       :: lmax, nmaps, lmin
       real(c_double), intent(in)  :: Dl(*), RJ2unit(*)
       real(c_double), intent(out) :: sqrtS_mat(*), sqrtInvS_mat(*), S_mat(*)
       integer(c_int)              :: ierr
     end function cmdr_cl_update_S

     ! sample_Dl_lookup (comm_Cl_mod.f90:1063-1145): one rand_uni(handle) in `uniform`; chosen = 0-based model index
     function cmdr_cl_sample_lookup(lmax, lmin_lookup, lmax_lookup, nmodel, Dl_lookup, active, sigma_l, S_mat, RJ2unit, &
          & uniform, Dl, chosen) bind(c, name='cmdr_cl_sample_lookup') result(ierr)
       import :: c_int, c_double
       integer(c_int), value         :: lmax, lmin_lookup, lmax_lookup, nmodel
       real(c_double), intent(in)    :: Dl_lookup(*), sigma_l(*), S_mat(*), RJ2unit(*)
       integer(c_int), intent(in)    :: active(6)
       real(c_double), value         :: uniform
       real(c_double), intent(inout) :: Dl(*)
       integer(c_int), intent(out)   :: chosen
       integer(c_int)                :: ierr
     end function cmdr_cl_sample_lookup

     ! get_Cl_apod folded into the tables handed to cmdr_comp_add / cmdr_comp_set_cl (comm_Cl_mod.f90:572-704)
     function cmdr_cl_apply_apod(lmax, nmaps, l_apod, lmax_prior, sqrtS_mat, sqrtInvS_mat, S_mat) &
          & bind(c, name='cmdr_cl_apply_apod') result(ierr)
       import :: c_int, c_double
       integer(c_int), value         :: lmax, nmaps, l_apod, lmax_prior
       real(c_double), intent(inout) :: sqrtS_mat(*), sqrtInvS_mat(*), S_mat(*)
       integer(c_int)                :: ierr
     end function cmdr_cl_apply_apod

     ! sample_Cls_inverse_wishart2 for cltype 'binned' (comm_Cl_mod.f90:1008-1249): one rand_uni(handle) per sampled
     ! bin goes in through `uniform`; returns 0, or 1 for ok = .false.
     function cmdr_cl_sample_binned(lmax, nmaps, sigma_l, S_mat, RJ2unit, nbin, bins, uniform, nuniform, Dl, nused) &
          & bind(c, name='cmdr_cl_sample_binned') result(ierr)
       import :: c_int, c_double, cmdr_cl_bin
       integer(c_int), value          :: lmax, nmaps, nbin, nuniform
       real(c_double), intent(in)     :: sigma_l(*), S_mat(*), RJ2unit(*), uniform(*)
       type(cmdr_cl_bin), intent(in)  :: bins(*)
       real(c_double), intent(inout)  :: Dl(*)
       integer(c_int), intent(out)    :: nused
       integer(c_int)                 :: ierr
     end function cmdr_cl_sample_binned

     ! compute_residual (comm_chisq_mod.f90:196-267) for every band; data / resid: arrays of c_loc(map) per band
     function cmdr_compute_residual(ctx, amp, data, resid) bind(c, name='cmdr_compute_residual') result(ierr)
       import :: c_int, c_ptr, c_double
       type(c_ptr),    value      :: ctx
       real(c_double), intent(in) :: amp(*)
       type(c_ptr),    intent(in) :: data(*), resid(*)
       integer(c_int)             :: ierr
     end function cmdr_compute_residual

     function cmdr_compute_residual_dev(ctx, amp_dev, data_dev, resid_dev) bind(c, name='cmdr_compute_residual_dev') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value      :: ctx, amp_dev
       type(c_ptr), intent(in) :: data_dev(*), resid_dev(*)
       integer(c_int)          :: ierr
     end function cmdr_compute_residual_dev

     ! applyMonoDipolePrior (comm_diffuse_comp_mod.f90:5738-5827) after cr_x2amp (comm_signal_mod.f90:186-194):
     ! prior_type 1 = 'monopole', 2 = 'monopole+dipole'; amp(ncr) stacked amplitudes, edited in place; b_l_out may be
     ! c_null_ptr (no output beam); mask = mono_prior_map%map(:,1) on the local pixels; mu(4) = the fit the reference prints
     function cmdr_apply_mono_dipole_prior(ctx, comp, amp, nside, b_l_out, mask, npix_local, prior_type, mu) &
          & bind(c, name='cmdr_apply_mono_dipole_prior') result(ierr)
       import :: c_int, c_ptr, c_double, c_int64_t
       type(c_ptr),        value         :: ctx, b_l_out
       integer(c_int),     value         :: comp, nside, prior_type
       integer(c_int64_t), value         :: npix_local
       real(c_double),     intent(inout) :: amp(*)
       real(c_double),     intent(in)    :: mask(*)
       real(c_double),     intent(out)   :: mu(4)
       integer(c_int)                    :: ierr
     end function cmdr_apply_mono_dipole_prior

     function cmdr_apply_mono_dipole_prior_dev(ctx, comp, amp_dev, nside, b_l_out, mask_dev, prior_type, mu) &
          & bind(c, name='cmdr_apply_mono_dipole_prior_dev') result(ierr)
       import :: c_int, c_ptr, c_double
       type(c_ptr),    value       :: ctx, amp_dev, b_l_out, mask_dev
       integer(c_int), value       :: comp, nside, prior_type
       real(c_double), intent(out) :: mu(4)
       integer(c_int)              :: ierr
     end function cmdr_apply_mono_dipole_prior_dev

     function cmdr_problem_info_ext(ctx, n, out) bind(c, name='cmdr_problem_info_ext') result(ierr)
       import :: c_int, c_ptr, c_int64_t
       type(c_ptr),        value       :: ctx
       integer(c_int),     value       :: n
       integer(c_int64_t), intent(out) :: out(*)
       integer(c_int)                  :: ierr
     end function cmdr_problem_info_ext

     function cmdr_matmulA(ctx, x, y) bind(c, name='cmdr_matmulA') result(ierr)
       import :: c_int, c_ptr, c_double
       type(c_ptr),    value       :: ctx
       real(c_double), intent(in)  :: x(*)
       real(c_double), intent(out) :: y(*)
       integer(c_int)              :: ierr
     end function cmdr_matmulA

     function cmdr_invM(ctx, x, y) bind(c, name='cmdr_invM') result(ierr)
       import :: c_int, c_ptr, c_double
       type(c_ptr),    value       :: ctx
       real(c_double), intent(in)  :: x(*)
       real(c_double), intent(out) :: y(*)
       integer(c_int)              :: ierr
     end function cmdr_invM

     function cmdr_compute_rhs(ctx, sample, resid, xi, eta, mu, rhs) bind(c, name='cmdr_compute_rhs') result(ierr)
       import :: c_int, c_ptr, c_double
       type(c_ptr),    value       :: ctx
       integer(c_int), value       :: sample
       type(c_ptr),    intent(in)  :: resid(*), xi(*)    ! c_loc(map%map) per band
       type(c_ptr),    value       :: eta, mu            ! c_loc(...) or c_null_ptr
       real(c_double), intent(out) :: rhs(*)
       integer(c_int)              :: ierr
     end function cmdr_compute_rhs

     function cmdr_solve(ctx, b, x, crit, tol, miniter, maxiter, check_freq, x0, niter, res, stat) &
          & bind(c, name='cmdr_solve') result(ierr)
       import :: c_int, c_ptr, c_double
       type(c_ptr),    value       :: ctx
       real(c_double), intent(in)  :: b(*)
       real(c_double), intent(out) :: x(*)
       integer(c_int), value       :: crit, miniter, maxiter, check_freq
       real(c_double), value       :: tol
       type(c_ptr),    value       :: x0                 ! c_null_ptr <=> cpar%cg_init_zero
       integer(c_int), intent(out) :: niter, stat
       real(c_double), intent(out) :: res(2)
       integer(c_int)              :: ierr
     end function cmdr_solve
     ! chain-file order of a_lm (float32, index l^2+l+m; comm_map_mod.f90:712-740) <-> packed columns
     function cmdr_alm_to_chain_order(alm, lmax, nmaps, chain32) bind(c, name='cmdr_alm_to_chain_order') result(ierr)
       import :: c_int, c_double, c_float
       real(c_double), intent(in)  :: alm(*)
       integer(c_int), value       :: lmax, nmaps
       real(c_float),  intent(out) :: chain32(*)
       integer(c_int)              :: ierr
     end function cmdr_alm_to_chain_order

     function cmdr_alm_from_chain_order(chain32, lmax, nmaps, alm) bind(c, name='cmdr_alm_from_chain_order') result(ierr)
       import :: c_int, c_double, c_float
       real(c_float),  intent(in)  :: chain32(*)
       integer(c_int), value       :: lmax, nmaps
       real(c_double), intent(out) :: alm(*)
       integer(c_int)              :: ierr
     end function cmdr_alm_from_chain_order

     ! compact components (templates, point sources): scalar amplitudes with Gaussian prior; one sparse matrix per band
     function cmdr_compact_add(ctx, nparam, sigma, mean, active) bind(c, name='cmdr_compact_add') result(idx)
       import :: c_int, c_ptr, c_double
       type(c_ptr),    value      :: ctx
       integer(c_int), value      :: nparam, active
       real(c_double), intent(in) :: sigma(*), mean(*)
       integer(c_int)             :: idx
     end function cmdr_compact_add

     function cmdr_compact_set_band(ctx, block, band, nnz, cell, param, val) bind(c, name='cmdr_compact_set_band') result(ierr)
       import :: c_int, c_int64_t, c_ptr, c_double
       type(c_ptr),        value      :: ctx
       integer(c_int),     value      :: block, band
       integer(c_int64_t), value      :: nnz
       integer(c_int64_t), intent(in) :: cell(*)     ! pix_local + npix_local * stokes, 0-based
       integer(c_int),     intent(in) :: param(*)    ! 0-based
       real(c_double),     intent(in) :: val(*)
       integer(c_int)                 :: ierr
     end function cmdr_compact_set_band

     ! comm_N_QUcov band: dense inverse covariance and its symmetric square root on the stacked (Q;U) pixels
     function cmdr_band_set_qucov(ctx, band, iN, siN_mat) bind(c, name='cmdr_band_set_qucov') result(ierr)
       import :: c_int, c_ptr, c_double
       type(c_ptr),    value      :: ctx
       integer(c_int), value      :: band
       real(c_double), intent(in) :: iN(*), siN_mat(*)
       integer(c_int)             :: ierr
     end function cmdr_band_set_qucov

     ! comm_map%getSigmaL (comm_map_mod.f90:1302-1351): alm(nalm, nmaps) -> sigma_l(0:lmax, nspec)
     function cmdr_sigma_l(alm, lmax, nmaps, sigma_l) bind(c, name='cmdr_sigma_l') result(ierr)
       import :: c_int, c_double
       real(c_double), intent(in)  :: alm(*)
       integer(c_int), value       :: lmax, nmaps
       real(c_double), intent(out) :: sigma_l(*)
       integer(c_int)              :: ierr
     end function cmdr_sigma_l

     ! ---- device selection and device-resident vectors (one MPI rank per GPU: cmdr_set_device(local rank) first)
     function cmdr_set_device(device) bind(c, name='cmdr_set_device') result(ierr)
       import :: c_int
       integer(c_int), value :: device
       integer(c_int)        :: ierr
     end function cmdr_set_device

     function cmdr_device_synchronize() bind(c, name='cmdr_device_synchronize') result(ierr)
       import :: c_int
       integer(c_int) :: ierr
     end function cmdr_device_synchronize

     function cmdr_dev_alloc(nbytes, ptr) bind(c, name='cmdr_dev_alloc') result(ierr)
       import :: c_int, c_size_t, c_ptr
       integer(c_size_t), value :: nbytes
       type(c_ptr), intent(out) :: ptr
       integer(c_int)           :: ierr
     end function cmdr_dev_alloc

     function cmdr_dev_free(ptr) bind(c, name='cmdr_dev_free') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ptr
       integer(c_int)     :: ierr
     end function cmdr_dev_free

     function cmdr_dev_mem_info(free_bytes, total_bytes) bind(c, name='cmdr_dev_mem_info') result(ierr)
       import :: c_int, c_size_t
       integer(c_size_t), intent(out) :: free_bytes, total_bytes
       integer(c_int)                 :: ierr
     end function cmdr_dev_mem_info

     function cmdr_memcpy_h2d(dst_dev, src_host, nbytes) bind(c, name='cmdr_memcpy_h2d') result(ierr)
       import :: c_int, c_size_t, c_ptr
       type(c_ptr), value       :: dst_dev, src_host      ! src_host = c_loc(array)
       integer(c_size_t), value :: nbytes
       integer(c_int)           :: ierr
     end function cmdr_memcpy_h2d

     function cmdr_memcpy_d2h(dst_host, src_dev, nbytes) bind(c, name='cmdr_memcpy_d2h') result(ierr)
       import :: c_int, c_size_t, c_ptr
       type(c_ptr), value       :: dst_host, src_dev
       integer(c_size_t), value :: nbytes
       integer(c_int)           :: ierr
     end function cmdr_memcpy_d2h

     ! page-lock a long-lived host array (c_loc(array)) so that the host-pointer entry points copy by DMA
     function cmdr_host_register(ptr_host, nbytes) bind(c, name='cmdr_host_register') result(ierr)
       import :: c_int, c_size_t, c_ptr
       type(c_ptr), value       :: ptr_host
       integer(c_size_t), value :: nbytes
       integer(c_int)           :: ierr
     end function cmdr_host_register

     function cmdr_host_unregister(ptr_host) bind(c, name='cmdr_host_unregister') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ptr_host
       integer(c_int)     :: ierr
     end function cmdr_host_unregister

     ! device-pointer forms of the CR entry points: x, b stay in HBM between calls (type(c_ptr) from cmdr_dev_alloc)
     function cmdr_matmulA_dev(ctx, x_dev, y_dev) bind(c, name='cmdr_matmulA_dev') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, x_dev, y_dev
       integer(c_int)     :: ierr
     end function cmdr_matmulA_dev

     function cmdr_invM_dev(ctx, x_dev, y_dev) bind(c, name='cmdr_invM_dev') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, x_dev, y_dev
       integer(c_int)     :: ierr
     end function cmdr_invM_dev

     function cmdr_compute_rhs_dev(ctx, sample, resid_dev, xi_dev, eta_dev, mu_dev, rhs_dev) &
          & bind(c, name='cmdr_compute_rhs_dev') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr),    value      :: ctx
       integer(c_int), value      :: sample
       type(c_ptr),    intent(in) :: resid_dev(*), xi_dev(*)     ! one device pointer per band
       type(c_ptr),    value      :: eta_dev, mu_dev, rhs_dev    ! eta / mu may be c_null_ptr
       integer(c_int)             :: ierr
     end function cmdr_compute_rhs_dev

     function cmdr_solve_dev(ctx, b_dev, x_dev, crit, tol, miniter, maxiter, check_freq, x0_dev, niter, res, stat) &
          & bind(c, name='cmdr_solve_dev') result(ierr)
       import :: c_int, c_ptr, c_double
       type(c_ptr),    value       :: ctx, b_dev, x_dev, x0_dev
       integer(c_int), value       :: crit, miniter, maxiter, check_freq
       real(c_double), value       :: tol
       integer(c_int), intent(out) :: niter, stat
       real(c_double), intent(out) :: res(2)
       integer(c_int)              :: ierr
     end function cmdr_solve_dev

     function cmdr_sigma_l_dev(alm_dev, stride, lmax, nmaps, sigma_l_dev) bind(c, name='cmdr_sigma_l_dev') result(ierr)
       import :: c_int, c_int64_t, c_ptr
       type(c_ptr),        value :: alm_dev, sigma_l_dev
       integer(c_int64_t), value :: stride
       integer(c_int),     value :: lmax, nmaps
       integer(c_int)            :: ierr
     end function cmdr_sigma_l_dev

     ! ---- accessors
     function cmdr_ctx_set_only_pol(ctx, only_pol) bind(c, name='cmdr_ctx_set_only_pol') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr),    value :: ctx
       integer(c_int), value :: only_pol           ! cpar%only_pol (comm_cr_mod.f90:600-601, 686-689)
       integer(c_int)        :: ierr
     end function cmdr_ctx_set_only_pol

     function cmdr_band_npix(ctx, band) bind(c, name='cmdr_band_npix') result(n)
       import :: c_int, c_int64_t, c_ptr
       type(c_ptr),    value :: ctx
       integer(c_int), value :: band
       integer(c_int64_t)    :: n
     end function cmdr_band_npix

     function cmdr_get_invN_diag(ctx, band, out_host) bind(c, name='cmdr_get_invN_diag') result(ierr)
       import :: c_int, c_ptr, c_double
       type(c_ptr),    value       :: ctx
       integer(c_int), value       :: band
       real(c_double), intent(out) :: out_host(*)  ! data(band)%N%invN_diag%alm (nalm, nmaps)
       integer(c_int)              :: ierr
     end function cmdr_get_invN_diag

     function cmdr_get_alpha_nu(ctx, band, out_host) bind(c, name='cmdr_get_alpha_nu') result(ierr)
       import :: c_int, c_ptr, c_double
       type(c_ptr),    value       :: ctx
       integer(c_int), value       :: band
       real(c_double), intent(out) :: out_host(*)  ! data(band)%N%alpha_nu(1:nmaps)
       integer(c_int)              :: ierr
     end function cmdr_get_alpha_nu

     function cmdr_problem_info(ctx, info) bind(c, name='cmdr_problem_info') result(ierr)
       import :: c_int, c_int64_t, c_ptr
       type(c_ptr),        value       :: ctx
       integer(c_int64_t), intent(out) :: info(3)
       integer(c_int)                  :: ierr
     end function cmdr_problem_info

     function cmdr_profile_enable(ctx, on) bind(c, name='cmdr_profile_enable') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr),    value :: ctx
       integer(c_int), value :: on
       integer(c_int)        :: ierr
     end function cmdr_profile_enable

     function cmdr_profile_read(ctx, ms_sum, count) bind(c, name='cmdr_profile_read') result(ierr)
       import :: c_int, c_ptr, c_double, c_long_long
       type(c_ptr), value            :: ctx
       real(c_double),       intent(out) :: ms_sum(4)   ! synthesis, ring stage, adjoint, whole matvec
       integer(c_long_long), intent(out) :: count(4)
       integer(c_int)                :: ierr
     end function cmdr_profile_read

     function cmdr_cl_apod(l, l_apod, lmax, lmax_prior, positive) bind(c, name='cmdr_cl_apod') result(f)
       import :: c_int, c_double
       integer(c_int), value :: l, l_apod, lmax, lmax_prior, positive
       real(c_double)        :: f
     end function cmdr_cl_apod

     ! ---- SHT level, the rest: polarised plans and the spin-2 call of exec_sharp_Y on columns 2:3
     function cmdr_sht_plan_create_pol(nside, lmax, nrings, rings, wring, max_maps, plan) &
          & bind(c, name='cmdr_sht_plan_create_pol') result(ierr)
       import :: c_int, c_ptr
       integer(c_int), value        :: nside, lmax, nrings, max_maps
       type(c_ptr),    value        :: rings, wring
       type(c_ptr),    intent(out)  :: plan
       integer(c_int)               :: ierr
     end function cmdr_sht_plan_create_pol

     function cmdr_sht_nalm(plan) bind(c, name='cmdr_sht_nalm') result(n)
       import :: c_int64_t, c_ptr
       type(c_ptr), value :: plan
       integer(c_int64_t) :: n
     end function cmdr_sht_nalm

     function cmdr_sht_npix(plan) bind(c, name='cmdr_sht_npix') result(n)
       import :: c_int64_t, c_ptr
       type(c_ptr), value :: plan
       integer(c_int64_t) :: n
     end function cmdr_sht_npix

     function cmdr_sht_execute_dev(plan, job, nmaps, alm_dev, alm_stride, map_dev, map_stride) &
          & bind(c, name='cmdr_sht_execute_dev') result(ierr)
       import :: c_int, c_int64_t, c_ptr
       type(c_ptr),        value :: plan, alm_dev, map_dev
       integer(c_int),     value :: job, nmaps
       integer(c_int64_t), value :: alm_stride, map_stride
       integer(c_int)            :: ierr
     end function cmdr_sht_execute_dev

     function cmdr_sht_execute_spin2(plan, job, almE, almB, mapQ, mapU) bind(c, name='cmdr_sht_execute_spin2') result(ierr)
       import :: c_int, c_ptr, c_double
       type(c_ptr),    value         :: plan
       integer(c_int), value         :: job
       real(c_double), intent(inout) :: almE(*), almB(*), mapQ(*), mapU(*)
       integer(c_int)                :: ierr
     end function cmdr_sht_execute_spin2

     function cmdr_sht_execute_spin2_dev(plan, job, almE_dev, almB_dev, mapQ_dev, mapU_dev) &
          & bind(c, name='cmdr_sht_execute_spin2_dev') result(ierr)
       import :: c_int, c_ptr
       type(c_ptr),    value :: plan, almE_dev, almB_dev, mapQ_dev, mapU_dev
       integer(c_int), value :: job
       integer(c_int)        :: ierr
     end function cmdr_sht_execute_spin2_dev

  end interface

contains

  subroutine cmdr_check(ierr, where)
    integer(c_int),   intent(in) :: ierr
    character(len=*), intent(in) :: where
    character(kind=c_char), pointer :: cmsg(:)
    integer :: i
    if (ierr >= 0) return
    call c_f_pointer(cmdr_last_error(), cmsg, [512])
    write(*,'(a)',advance='no') 'cmdr_hip error in '//trim(where)//': '
    do i = 1, 512
       if (cmsg(i) == c_null_char) exit
       write(*,'(a)',advance='no') cmsg(i)
    end do
    write(*,*)
    stop 1
  end subroutine cmdr_check

end module cmdr_hip_mod
