// The reference's backend test (totsu_f32cuda/tests/solver.rs:16-54 == totsu_core/tests/solver.rs:14-53) on F32HIP.
// AUTHORED, NOT COMPILED; the same known-answer test runs on the GPU through tests/test_gpu_solver.py::test_kat_psd.
use float_eq::assert_float_eq;
use totsu_core::solver::{Operator, Solver};
use totsu_core::{ConePSD, MatOp, MatType};
use totsu_f32hip::F32HIP;

type La = F32HIP;

#[test]
fn test_solver() {
    totsu_f32hip::init(0);
    let op_c = MatOp::<La>::new(MatType::General(1, 1), &[1.]);
    let op_a = MatOp::<La>::new(MatType::General(3, 1), &[0., -1. * 1.41421356, -3.]);
    let op_b = MatOp::<La>::new(MatType::General(3, 1), &[1., 0. * 1.41421356, 10.]);
    let s = Solver::<La>::new().par(|p| { p.max_iter = Some(100_000); p.eps_acc = 1e-5; });
    let mut cone_w = vec![0.; ConePSD::<La>::query_worklen(op_a.size().0)];
    let cone = ConePSD::<La>::new(&mut cone_w, s.par.eps_zero);
    let mut work = vec![0.; Solver::<La>::query_worklen(op_a.size())];
    let rslt = s.solve((op_c, op_a, op_b, cone, &mut work)).unwrap();
    assert_float_eq!(rslt.0[0], -2., abs_all <= 1e-3);
}
