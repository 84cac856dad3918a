// c2_dispatch.hpp -- the ONE table of everything that steers the dispatch of this library: switches that force or
// disable an alternative formulation (every alternative is parity-tested, so a switch changes speed, never results beyond
// rounding) and the thresholds / cost-model constants the automatic choices use, each with the measurement it came from.
//
// Values are read from the environment ONCE, when the library is loaded; after that only c2_set_option() changes them
// (include/celerite2_amd.h).  No entry point calls getenv: a workspace-size query and the call that follows it see the same
// options unless the caller changes them in between.  INTEGRATION.md section 5 is generated from this table
// (tools/gen_dispatch_doc.py); tools/crossovers.py re-measures the crossovers on the box it runs on and prints the
// c2_set_option() calls that would move them.
//
// X(id, environment variable, default, kind, what it does, where the default comes from)
//   kind 's': switch -- unset means "automatic"; has() tells whether it was set
//   kind 't': threshold / constant -- val() is the default unless set
#pragma once

#define C2_OPTIONS(X)                                                                                                                 \
  X(lanes, "C2_LANES", 0, 's', "lane mapping of the fused log-likelihood kernels: 8 (a group of lanes per series), 4 (four lanes per series, two columns per lane, J = 8), 2 (two lanes per series, J = 8), 1 (one lane per series); unset: by batch size", "profiles/r02_lane_mappings.md") \
  X(lanes1_min_batch_fwd, "C2_LANES1_MIN_BATCH_FWD", 24576, 't', "forward log-likelihood: one lane per series from this many series up (widths 8, 6, 4, 2)", "N = 4096, J = 8: 3.3 vs 4.1 ms at 24576 series, 6.7 vs 10.4 ms at 65536 (profiles/r02_lane_mappings.md)") \
  X(lanes1_min_batch_grad, "C2_LANES1_MIN_BATCH_GRAD", 24576, 't', "log-likelihood + gradient: one lane per series from this many series up (widths 8, 4, 2)", "15.7 vs 16.0 ms at 24576 series, 28.2 vs 41.8 ms at 65536 (profiles/r02_lane_mappings.md)") \
  X(lanes1_min_batch_grad_j6, "C2_LANES1_MIN_BATCH_GRAD_J6", 32768, 't', "the same at width 6 (rows of 48 bytes: no aligned 128-byte runs)", "18.5 vs 20.7 ms at 32768 series, 17.2 vs 15.8 ms at 24576") \
  X(lanes2_min_batch_fwd, "C2_LANES2_MIN_BATCH_FWD", 16385, 't', "forward log-likelihood, J = 8: two lanes per series from this many series up ...", "N = 4096: 2.38 - 2.57 vs 2.25 - 2.43 ms at 16384 series (two columns per lane: one wavefront per SIMD there), 2.59 vs 3.69 at 20480 (one lane), 2.76 vs 3.70 at 24576, 3.37 vs 3.84 at 32768 (profiles/r04_two_lanes.md)") \
  X(lanes2_min_batch_grad, "C2_LANES2_MIN_BATCH_GRAD", 16385, 't', "log-likelihood + gradient, J = 8: two lanes per series from this many series up ...", "11.0 vs 14.6 ms at 18432 series (8 lanes: a third round of wavefronts), 11.4 vs 14.4 at 20480, 12.2 - 12.5 vs 15.2 at 24576; 10.25 vs 10.11 at 16384, 10.17 vs 9.65 at 14336 (8 lanes, reverse sweep by the backward recursion; profiles/r04_two_lanes.md)") \
  X(lanes2_max_batch, "C2_LANES2_MAX_BATCH", 32768, 't', "... up to this many (32 series per wavefront: one wavefront per SIMD)", "14.9 - 16.0 vs 16.2 - 17.2 ms at 32768 series (box to box), 27.5 vs 19.4 at 34816 (profiles/r04_two_lanes.md)") \
  X(loglik_back, "C2_LOGLIK_BACK", 1, 's', "log-likelihood + gradient on the group mappings (up to eight lanes per series): reverse sweep by the BACKWARD recursion from recorded W rows instead of replaying the forward steps; 0 keeps the replay (A/B runs)", "profiles/r04_back8.md") \
  X(loglik_back_occ2, "C2_LOGLIK_BACK_OCC2", 1, 's', "... for batches with more wavefronts than the chip has SIMDs (J = 8: 8192 < B <= 16384) as instances that fit two wavefronts per SIMD; 0: one per SIMD, the rest of the batch behind the first part (A/B runs)", "profiles/r04_back8.md") \
  X(loglik_scaled, "C2_LOGLIK_SCALED", 1, 's', "... in a scaled frame (states multiplied by exp(-c (t_anchor - t_n)): no decay factors in the step, three gathered vectors instead of five); 0: the plain backward recursion (A/B runs)", "profiles/r05_scaled_frame.md") \
  X(loglik_lines, "C2_LOGLIK_LINES", 0, 's', "1: the eight-lane log-likelihood kernels (J = 8) request the rows of U, V (and store bU, bV) as aligned 128-byte lines through an LDS tile instead of one 64-byte row at a time; 2: in the forward pass only, 3: in the reverse sweep only (A/B runs; off by default)", "N = 4096: 8192 series 4.69 vs 4.72 ms, 1024 series 3.49 vs 3.20, 4096 series 3.72 vs 3.33 -- the LDS detour costs a lone wavefront what the address unit gives back (profiles/r05_lines.md)") \
  X(fwd_dpp_gathers, "C2_FWD_DPP_GATHERS", 0, 's', "1: forward kernels of the group mappings (up to eight lanes per series) gather the next step's decay and U vectors by DPP permutes instead of through LDS (A/B runs)", "slower: 1024 series x 4096 rows 3.76 -> 4.11 ms for the gradient pair (profiles/r04_back8.md)") \
  X(lanes4_min_batch, "C2_LANES4_MIN_BATCH", 16384, 't', "forward log-likelihood, J = 8: two columns per lane from this many series up", "14-15 % faster from 16384 series, equal at 8192 (profiles/r01_lanes4.md)") \
  X(lanes4_min_batch_grad, "C2_LANES4_MIN_BATCH_GRAD", 9216, 't', "log-likelihood + gradient, J = 8: four lanes per series (two columns per lane, scaled frame; c2_loglik_q4.hip) from this many series up ...", "N = 4096: 4.99 vs 4.52 ms at 8192 series (8 lanes: one wavefront per SIMD there), 6.10 vs 6.43 at 9216, 6.21 vs 6.60 at 10240, 6.94 vs 7.36 at 12288 (profiles/r05_four_lanes.md)") \
  X(lanes4_max_batch_grad, "C2_LANES4_MAX_BATCH_GRAD", 16384, 't', "... up to this many (one wavefront per SIMD; the two-lane pair takes over beyond)", "8.39 ms at 16384 series against 9.7 on eight lanes at two wavefronts per SIMD (profiles/r05_four_lanes.md)") \
  X(loglik_q4_lines, "C2_LOGLIK_Q4_LINES", 0, 's', "the four-lane gradient pair moves the rows of U, V, bU, bV as aligned 128-byte lines of eight series through LDS tiles (1) or 64 bytes of sixteen series at a time (0); unset: lines from C2_Q4_LINES_MIN_BATCH series", "profiles/r05_four_lanes.md") \
  X(q4_lines_min_batch, "C2_Q4_LINES_MIN_BATCH", 11264, 't', "... that batch size", "N = 4096, lines vs rows: 6.53 vs 6.55 ms at 10240 series, 6.68 vs 6.87 at 12288, 8.20 vs 9.00 at 14336, 8.39 vs 8.96 at 16384; 5.53 vs 5.29 at 8192 (profiles/r05_four_lanes.md)") \
  X(timepar, "C2_TIMEPAR", 0, 's', "forward log-likelihood / factor (widths 4, 2) and the single-rhs solves parallel along TIME: 1 forces, 0 disables; unset: by a cost model (C2_TIMEPAR_ELEMENTS_BIAS) / for small batches of long series", "tools/onepass_grid.py, profiles/r05_onepass.md") \
  X(timepar_min_rows, "C2_TIMEPAR_MIN_ROWS", 1536, 't', "shortest series the time-parallel single-rhs SOLVES take when the batch is not a handful (B * J > 512)", "a handful of series from 384 / 704 / 1024 rows at widths 2 / 4 / 8 (profiles/r02_timepar.md)") \
  X(timepar_max_batch_x_width, "C2_TIMEPAR_MAX_BATCH_X_WIDTH", 8192, 't', "largest B * J the time-parallel single-rhs solves take", "linear in the batch beyond one wavefront per SIMD (profiles/r02_timepar.md)") \
  X(timepar_elements_bias, "C2_TIMEPAR_ELEMENTS_BIAS", 100, 't', "forward log-likelihood and factor at widths 4 and 2: the time-parallel forms on chunk elements are taken when their modelled time x this / 100 is below the row-by-row kernel's (the model: use_timepar in c2_loglik.hip; 50 favours them, 200 the row-by-row kernels)", "width 4, 1024 series: log-likelihood 0.044 ms up to 1024 rows, 0.109 at 4096 (row by row 0.22 us per row); factor 0.09 / 0.25 ms; 4096 x 4096: 0.38 vs 0.88 and 0.82 vs 1.02 ms; 8192 x 1024: 0.28 vs 0.24 (tools/onepass_grid.py, profiles/r05_onepass.md)") \
  X(timepar8_min_rows, "C2_TIMEPAR8_MIN_ROWS", 512, 't', "forward log-likelihood at width 8: shortest series the chunk elements (combined by workgroups, k_e8_tree) take", "profiles/r05_onepass.md") \
  X(timepar8_max_chunks, "C2_TIMEPAR8_MAX_CHUNKS", 32768, 't', "... and the largest number of 64-row chunks (B * ceil(N / 64))", "profiles/r05_onepass.md") \
  X(timepar_grad, "C2_TIMEPAR_GRAD", 0, 's', "log-likelihood GRADIENT (and factor_rev) parallel along time, widths 1 .. 8: 1 forces, 0 disables; unset: small batches of long series", "tools/timepar_grad_time.py, profiles/r02_timepar_grad.md") \
  X(timepar_grad_min_rows, "C2_TIMEPAR_GRAD_MIN_ROWS", 1024, 't', "shortest series the time-parallel gradient takes at widths 7, 8 beyond a handful of series (768 at widths 3 .. 6, 512 at 1, 2; 256 for at most 4096 chunks)", "one series draws level at ~400 / ~600 / ~800 rows at J = 2 / 4, 6 / 8 (tools/timepar_grad_time.py)") \
  X(timepar_grad_min_rows_handful, "C2_TIMEPAR_GRAD_MIN_ROWS_HANDFUL", 384, 't', "shortest series the time-parallel gradient takes for a handful of series (at most 4096 chunks of 64 rows; they run with 16-row chunks)", "round 5 (the row-by-row pair in the scaled frame), one series, J = 8: 0.336 vs 0.276 ms row by row at 256 rows, 0.381 vs 0.376 at 384, 0.430 vs 0.476 at 512, 0.87 vs 3.25 at 4096 (tools/crossovers.py)") \
  X(timepar_grad_max_chunks, "C2_TIMEPAR_GRAD_MAX_CHUNKS", 32768, 't', "largest number of 64-row chunks (B * ceil(N / 64)) the time-parallel gradient takes", "1024 x 4096 at J = 4: 3.6 vs 3.0 ms row by row") \
  X(timepar_cond_limit, "C2_TIMEPAR_COND_LIMIT", 0, 't', "if > 0: largest conditioning kappa = max a_n / d_n for which the result of the time-parallel gradient stands; beyond it the row-by-row kernels recompute the batch behind the device-side gate (0, the default: no limit)", "rounding moves ANY float64 evaluation by c eps kappa^2 of the largest gradient entry: the oracle itself c = 0.4 (against its own extended-precision evaluation), the row-by-row kernels ~0.05, the time-parallel form ~0.01 and 0.6 in the worst draw of 9000 -- typically the closer of the two to the oracle, hence no limit by default (tools/kappa_sweep.py, tools/verify_words.py, profiles/r03_timepar_verification.md)") \
  X(verify_fallback, "C2_VERIFY_FALLBACK", 1, 's', "0 (diagnostics only): keep the result of a time-parallel form whatever its device-side verification says", "tools/verify_words.py") \
  X(factor_iter, "C2_FACTOR_ITER", 0, 's', "factor by Newton iterations on the chunk start states: 1 forces, 0 disables; unset: from 2048 rows and at most 32768 chunks (widths 4, 2: from 32768 / 131072 rows)", "tools/factor_iter_time.py: 4096 rows 1.21 -> 0.76 ms, 1e5 rows 29.5 -> 1.0 ms at J = 8") \
  X(tpg_rows, "C2_TPG_ROWS", 0, 's', "chunk length of the time-parallel gradient / Newton factor / chunk-map solves: 16, 32 or 64; unset: 16 up to 4096 rows and 32 beyond for at most 4096 chunks of 64 rows, else 64", "one series of 4096 rows 1.33 -> 1.03 -> 0.73 ms (64 -> 32 -> 16 rows), 1e5 rows 1.68 -> 1.25 ms with 32") \
  X(tpg_rows16_max_rows, "C2_TPG_ROWS16_MAX_ROWS", 4096, 't', "longest series that takes 16-row chunks", "chains of more than 256 chunks cost accuracy first (9000 rows: 1.3e-11 vs 4e-12), then time") \
  X(dropin_long_rows, "C2_DROPIN_LONG_ROWS", 512, 't', "shortest series factor + S and factor_rev take in their time-parallel form (at least 128)", "tools/bench_ops.py, J = 8: factor_rev 1 x 512 0.38 -> 0.15 ms, 512 x 4096 3.0 -> 1.6 ms") \
  X(long_min_rows, "C2_LONG_MIN_ROWS", 512, 't', "shortest series the chunk-map solves (with F / several right-hand sides) and the long-series reverse sweeps take (at least 128)", "1 x 4096 + F 0.60 -> 0.06 ms, 64 x 1024 0.13 -> 0.05 ms (tools/bench_ops.py)") \
  X(rev_long, "C2_REV_LONG", 0, 's', "the four reverse sweeps as opposite sweep + per-row pass: 1 forces, 0 disables; unset: small batches of long series", "one series of 1e5 rows, J = 8: solve_lower_rev 18.3 -> 0.15 ms (profiles/r02_long_series_rocprof.md)") \
  X(scan_min_rows, "C2_SCAN_MIN_ROWS", 1024, 't', "shortest series the chunked products (c2_scan.hip) take on batches of at most 128 series (at least 256)", "1 x 4096 0.60 -> 0.08 ms, 64 x 4096 0.63 -> 0.15 ms; 1024 x 2048 would lose (0.36 -> 1.01 ms)") \
  X(scan_min_chunk, "C2_SCAN_MIN_CHUNK", 0, 's', "chunk length of the chunked products (at least 64); unset: the power of two next to 0.5 sqrt(N)", "one series, J = 8: 20000 rows 0.79 -> 0.16 ms, 1e5 rows 0.80 -> 0.32 ms") \
  X(solve_chunk_col_ms, "C2_SOLVE_CHUNK_COL_MS", 0.05, 't', "cost model of the chunk-map solves: fixed cost (ms) of one right-hand side", "64 x 4096: 0.07 ms, 512 x 4096: 0.24 ms (tools/bench_ops.py)") \
  X(solve_chunk_ms, "C2_SOLVE_CHUNK_MS", 6e-6, 't', "... cost (ms) per 64-row chunk and right-hand side", "same measurement") \
  X(solve_row_us, "C2_SOLVE_ROW_US", 0.15, 't', "... against the row-by-row sweep: microseconds per row whatever the batch", "4096 rows: 0.63 ms with one right-hand side") \
  X(solve_row_rhs_us, "C2_SOLVE_ROW_RHS_US", 0.025, 't', "... plus microseconds per row and right-hand side", "4096 rows: 1.29 ms with 8 right-hand sides") \
  X(sweep_cols, "C2_SWEEP_COLS", 1, 's', "0: forward sweeps with 9 .. 32 right-hand sides at J = 8 (no workspace) on the lanes-over-rhs kernel (16 / 32 lanes per series) instead of eight lanes per series with several columns per lane and Y / Z in groups of four rows", "B = 8192, N = 4096: profiles/r04_nrhs_scan.md") \
  X(solve_cols, "C2_SOLVE_COLS", 0, 's', "solve_lower / solve_upper without workspace as chunk maps with lanes over the right-hand sides (every column in the same three launches; J <= 16): 1 forces, 0 disables; unset: 16 and more right-hand sides on small batches, by the cost model of solve_cols_shape", "1 x 4096 with 1024 right-hand sides: profiles/r04_large_nrhs.md") \
  X(solve_cols_min_rows, "C2_SOLVE_COLS_MIN_ROWS", 512, 't', "shortest series that form takes", "below, a launch is a few hundred steps of 0.26 us: nothing to gain behind three launches") \
  X(mfma, "C2_MFMA", 1, 's', "0: long-series products (J = 16; 16 / 32 / 64 right-hand sides) on the VALU instead of the fp64 matrix cores", "N = 1e7: 3.1 vs 9.2 ms (profiles/r02_ubench_memory_and_mfma.md)") \
  X(general_tile, "C2_GENERAL_TILE", 1, 's', "0: general_matmul_* on the two-phase kernels instead of the row tiles", "B = 8192, N = M = 4096, nrhs = 1: 1.4 vs 4.05 ms (profiles/r02_general_matmul.md)") \
  X(generalk, "C2_GENERALK", 1, 's', "0: general_matmul_* with five or more right-hand sides on the first-round kernels", "nrhs = 8: 7.0 vs 24.3 ms") \
  X(general_tile_max_rhs, "C2_GENERAL_TILE_MAX_RHS", 4, 't', "general_matmul_* on batches of 512 series and more: up to this many right-hand sides by 64-row tiles (one pass per tile of 4), beyond it lanes over the right-hand sides", "profiles/r04_general_tile_rhs.md") \
  X(general_chunks, "C2_GENERAL_CHUNKS", 1, 's', "0: never cut general_matmul_* on small batches of long series into chunks", "B = 1, N = M = 1e5: 32 -> 0.155 ms") \
  X(sweepk_rev, "C2_SWEEPK_REV", 1, 's', "0: multi-rhs reverse sweeps on the lanes-over-J kernel instead of lanes over the right-hand sides", "nrhs = 8: 9.1 vs 21.5 ms") \
  X(sweep1_lines, "C2_SWEEP1_LINES", 1, 's', "0: the single-rhs sweeps at J = 8 request their two rows per step one by one instead of by aligned 128-byte lines", "B = 8192, N = 4096: profiles/r03_sweep_rev_lines.md") \
  X(sweepk_lines, "C2_SWEEPK_LINES", 1, 's', "0: the forward sweeps with nrhs = J = 8 on full wavefronts row by row instead of by aligned 128-byte lines", "B = 8192, N = 4096: profiles/r03_sweep_rev_lines.md") \
  X(s_replay_lines, "C2_S_REPLAY_LINES", 1, 's', "0: the S rows of factor-with-workspace at J = 8 replayed with one request per row for t, d and W instead of transposed tiles and aligned 128-byte lines", "B = 8192, N = 4096: profiles/r03_sweep_rev_lines.md") \
  X(sweept, "C2_SWEEPT", 1, 's', "0: forward sweeps with two to five right-hand sides (two or three with the workspace) on the lanes-over-rhs kernel (idle lanes) / the first-round kernel instead of lanes over J with transposed scalar streams", "B = 8192, N = 4096: profiles/r03_nrhs_scan.md") \
  X(sweept_rev, "C2_SWEEPT_REV", 1, 's', "0: reverse sweeps with two to four right-hand sides on the first-round kernel (every per-series scalar fetched per step) instead of transposed scalar streams", "B = 8192, N = 4096, nrhs = 3: profiles/r03_per_op_B8192.md") \
  X(sweep1_rev_lines, "C2_SWEEP1_REV_LINES", 1, 's', "0: the single-rhs reverse sweeps at J = 8 request and store their five rows per step one by one instead of by aligned 128-byte lines", "B = 8192, N = 4096: profiles/r03_sweep_rev_lines.md") \
  X(sweep_rev_lines, "C2_SWEEP_REV_LINES", 1, 's', "0: the reverse sweeps with nrhs = J = 8 on full wavefronts row by row instead of by aligned 128-byte lines", "B = 8192, N = 4096: measured in profiles/r03_sweep_rev_lines.md") \
  X(terms_fused, "C2_TERMS_FUSED", 0, 's', "coefficient-level log-likelihood: 1 forces the fused one-lane kernels (J = 8, 4, 2), 0 the composed chain; unset: by batch size", "65536 series: 21.1 ms fused; 8192 series: 9.2 ms composed") \
  X(terms_fused_min_batch_fwd, "C2_TERMS_FUSED_MIN_BATCH_FWD", 10240, 't', "coefficient-level forward: fused kernels from this many series up", "N = 4096, J = 8: fused 3.93 ms whatever the batch up to 32768; composed 3.52 at 8192 series, 4.47 at 12288, 5.39 at 16384 (tools/terms_time.py)") \
  X(terms_fused_min_batch_grad, "C2_TERMS_FUSED_MIN_BATCH_GRAD", 16384, 't', "coefficient-level gradient: fused kernels from this many series up", "N = 4096, J = 8: fused 16.0 - 16.5 ms whatever the batch up to 32768; composed 8.45 at 8192 series, 13.3 at 12288, 16.4 at 16384, 20.5 at 24576 (tools/terms_time.py)") \
  X(kron_banded, "C2_KRON_BANDED", 1, 's', "0: the per-band passes of the 2-D collapsed method with a thread per epoch for every M", "1.66 + 2.31 -> 0.30 + 0.15 ms at 32 x 50000 x 16")

namespace c2 {
namespace opt {
enum Id {
#define C2_OPT_ENUM(id, env, def, kind, doc, src) k_##id,
  C2_OPTIONS(C2_OPT_ENUM)
#undef C2_OPT_ENUM
      kCount
};
bool has(Id id);     // set explicitly: by the environment at load time or by c2_set_option
double val(Id id);   // its value if set, the table's default otherwise
inline long long ival(Id id) { return (long long)val(id); }
}  // namespace opt
}  // namespace c2
