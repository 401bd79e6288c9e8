// A known-answer LP through `Solver<F32HIP>` with totsu's unchanged `ProbLP` builder.  AUTHORED, NOT COMPILED (no cargo
// in the build environment); the GPU parity tests of this repository run through the same C ABI (tests/test_gpu_*.py).
//
//   minimise  -x0 - 2 x1      subject to   0 <= x0 <= 3,   0 <= x1 <= 1.5,   x0 + x1 <= 4
//   optimum at the vertex (2.5, 1.5), objective -5.5
use float_eq::assert_float_eq;
use totsu::prelude::*;
use totsu::*;
use totsu_f32hip::F32HIP;

type La = F32HIP;

#[test]
fn box_lp_reaches_the_vertex() {
    totsu_f32hip::init(0);
    let n = 2;
    let rows: [[f32; 2]; 5] = [[-1., 0.], [0., -1.], [1., 0.], [0., 1.], [1., 1.]];
    let rhs: [f32; 5] = [0., 0., 3., 1.5, 4.];

    let vec_c = MatBuild::<La>::new(MatType::General(n, 1)).iter_colmaj(&[-1f32, -2.]);
    let mat_g = MatBuild::<La>::new(MatType::General(rows.len(), n)).by_fn(|r, c| rows[r][c]);
    let vec_h = MatBuild::<La>::new(MatType::General(rhs.len(), 1)).iter_colmaj(&rhs);
    let mat_a = MatBuild::<La>::new(MatType::General(0, n));
    let vec_b = MatBuild::<La>::new(MatType::General(0, 1));

    let s = Solver::<La>::new().par(|p| { p.eps_acc = 1e-4; p.max_iter = Some(200_000); });
    let mut lp = ProbLP::<La>::new(vec_c, mat_g, vec_h, mat_a, vec_b);
    let (x, _y) = s.solve(lp.problem()).unwrap();
    assert_float_eq!(x[0..2], [2.5f32, 1.5].as_ref(), abs_all <= 2e-3);
}
