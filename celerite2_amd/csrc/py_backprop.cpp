// py_backprop.cpp -- `celerite2_amd.backprop`: drop-in for the reference's
// `celerite2.backprop` pybind11 module (python/celerite2/backprop.cpp:12-926):
// 7 *_fwd ops that also emit the autodiff workspaces S (N,J,J) / F (N,J,nrhs)
// and 5 *_rev ops.  Signature rule (python/spec/generate.py:21-30):
//   fwd(inputs..., outputs..., extra_outputs...) -> (outputs..., extra_outputs...)
//   rev(inputs..., outputs..., extra_outputs..., b_outputs..., b_inputs...) -> (b_inputs...)
#define C2PY_LINALG_EXCEPTION backprop_linalg_exception
#include "py_common.hpp"

using namespace c2py;

namespace {

// backprop.factor_fwd -- backprop.cpp:12-67
auto factor_fwd(Arr t, Arr c, Arr a, Arr U, Arr V, Arr d, Arr W, Arr S) {
  py::buffer_info tb = t.request(), cb = c.request(), ab = a.request(), Ub = U.request(), Vb = V.request(),
                  db = d.request(), Wb = W.request(), Sb = S.request();
  const py::ssize_t N = dim0(tb, "t"), J = dim0(cb, "c");
  want1(tb, N, "t"); want1(cb, J, "c"); want1(ab, N, "a"); want2(Ub, N, J, "U"); want2(Vb, N, J, "V");
  want1(db, N, "d"); want2(Wb, N, J, "W"); want3(Sb, N, J, J, "S");
  int64_t flag = 0;
  int rc;
  {
    py::gil_scoped_release nogil;
    rc = c2h_factor(N, J, cptr(tb), cptr(cb), cptr(ab), cptr(Ub), cptr(Vb), mptr(db), mptr(Wb), mptr(Sb), &flag);
  }
  check(rc);
  if (flag) throw linalg_exception();
  return std::make_tuple(d, W, S);
}

// backprop.factor_rev -- backprop.cpp:68-150.  (The S workspace is validated and shipped like every argument; on series of 512
// rows and more the device replays its states from d, W instead and reads S only if the device-side verification of that
// time-parallel pass fails -- same result as the reference on the same inputs, see celerite2_amd/ops.py: factor_rev.)
auto factor_rev(Arr t, Arr c, Arr a, Arr U, Arr V, Arr d, Arr W, Arr S, Arr bd, Arr bW, Arr bt, Arr bc, Arr ba, Arr bU,
                Arr bV) {
  py::buffer_info tb = t.request(), cb = c.request(), ab = a.request(), Ub = U.request(), Vb = V.request(),
                  db = d.request(), Wb = W.request(), Sb = S.request(), bdb = bd.request(), bWb = bW.request(),
                  btb = bt.request(), bcb = bc.request(), bab = ba.request(), bUb = bU.request(), bVb = bV.request();
  const py::ssize_t N = dim0(tb, "t"), J = dim0(cb, "c");
  want1(tb, N, "t"); want1(cb, J, "c"); want1(ab, N, "a"); want2(Ub, N, J, "U"); want2(Vb, N, J, "V");
  want1(db, N, "d"); want2(Wb, N, J, "W"); want3(Sb, N, J, J, "S"); want1(bdb, N, "bd"); want2(bWb, N, J, "bW");
  want1(btb, N, "bt"); want1(bcb, J, "bc"); want1(bab, N, "ba"); want2(bUb, N, J, "bU"); want2(bVb, N, J, "bV");
  int rc;
  {
    py::gil_scoped_release nogil;
    rc = c2h_factor_rev(N, J, cptr(tb), cptr(cb), cptr(ab), cptr(Ub), cptr(Vb), cptr(db), cptr(Wb), cptr(Sb),
                        cptr(bdb), cptr(bWb), mptr(btb), mptr(bcb), mptr(bab), mptr(bUb), mptr(bVb));
  }
  check(rc);
  return std::make_tuple(bt, bc, ba, bU, bV);
}

// backprop.{solve_lower,solve_upper,matmul_lower,matmul_upper}_fwd -- backprop.cpp:153-215, 305-367, 457-519, 609-671
// (Z is zeroed first, e.g. backprop.cpp:201,207; for solves the op then sets Z = Y.)
template <int OP>
auto sweep_fwd(Arr t, Arr c, Arr U, Arr W, Arr Y, Arr Z, Arr F) {
  py::buffer_info tb = t.request(), cb = c.request(), Ub = U.request(), Wb = W.request(), Yb = Y.request(),
                  Zb = Z.request(), Fb = F.request();
  const py::ssize_t N = dim0(tb, "t"), J = dim0(cb, "c"), nrhs = dim1(Yb, "Y");
  want1(tb, N, "t"); want1(cb, J, "c"); want2(Ub, N, J, "U"); want2(Wb, N, J, OP < 2 ? "W" : "V");
  want2(Yb, N, nrhs, "Y"); want2(Zb, N, nrhs, "Z"); want3(Fb, N, J, nrhs, "F");
  int rc;
  {
    py::gil_scoped_release nogil;
    if (OP == 0) rc = c2h_solve_lower(N, J, nrhs, cptr(tb), cptr(cb), cptr(Ub), cptr(Wb), cptr(Yb), mptr(Zb), mptr(Fb));
    else if (OP == 1) rc = c2h_solve_upper(N, J, nrhs, cptr(tb), cptr(cb), cptr(Ub), cptr(Wb), cptr(Yb), mptr(Zb), mptr(Fb));
    else if (OP == 2) rc = c2h_matmul_lower(N, J, nrhs, cptr(tb), cptr(cb), cptr(Ub), cptr(Wb), cptr(Yb), mptr(Zb), mptr(Fb), 1);
    else rc = c2h_matmul_upper(N, J, nrhs, cptr(tb), cptr(cb), cptr(Ub), cptr(Wb), cptr(Yb), mptr(Zb), mptr(Fb), 1);
  }
  check(rc);
  return std::make_tuple(Z, F);
}

// backprop.{...}_rev -- backprop.cpp:216-302, 368-454, 520-606, 672-758
template <int OP>
auto sweep_rev(Arr t, Arr c, Arr U, Arr W, Arr Y, Arr Z, Arr F, Arr bZ, Arr bt, Arr bc, Arr bU, Arr bW, Arr bY) {
  py::buffer_info tb = t.request(), cb = c.request(), Ub = U.request(), Wb = W.request(), Yb = Y.request(),
                  Zb = Z.request(), Fb = F.request(), bZb = bZ.request(), btb = bt.request(), bcb = bc.request(),
                  bUb = bU.request(), bWb = bW.request(), bYb = bY.request();
  const py::ssize_t N = dim0(tb, "t"), J = dim0(cb, "c"), nrhs = dim1(Yb, "Y");
  want1(tb, N, "t"); want1(cb, J, "c"); want2(Ub, N, J, "U"); want2(Wb, N, J, OP < 2 ? "W" : "V");
  want2(Yb, N, nrhs, "Y"); want2(Zb, N, nrhs, "Z"); want3(Fb, N, J, nrhs, "F"); want2(bZb, N, nrhs, "bZ");
  want1(btb, N, "bt"); want1(bcb, J, "bc"); want2(bUb, N, J, "bU"); want2(bWb, N, J, OP < 2 ? "bW" : "bV");
  want2(bYb, N, nrhs, "bY");
  int rc;
  {
    py::gil_scoped_release nogil;
    auto fn = OP == 0 ? c2h_solve_lower_rev : OP == 1 ? c2h_solve_upper_rev : OP == 2 ? c2h_matmul_lower_rev : c2h_matmul_upper_rev;
    rc = fn(N, J, nrhs, cptr(tb), cptr(cb), cptr(Ub), cptr(Wb), cptr(Yb), cptr(Zb), cptr(Fb), cptr(bZb), mptr(btb),
            mptr(bcb), mptr(bUb), mptr(bWb), mptr(bYb));
  }
  check(rc);
  return std::make_tuple(bt, bc, bU, bW, bY);
}

// backprop.general_matmul_{lower,upper}_fwd -- backprop.cpp:761-901 (Z zeroed, F (M,J,nrhs); no _rev)
template <bool LOWER>
auto general_fwd(Arr t1, Arr t2, Arr c, Arr U, Arr V, Arr Y, Arr Z, Arr F) {
  py::buffer_info t1b = t1.request(), t2b = t2.request(), cb = c.request(), Ub = U.request(), Vb = V.request(),
                  Yb = Y.request(), Zb = Z.request(), Fb = F.request();
  const py::ssize_t N = dim0(t1b, "t1"), M = dim0(t2b, "t2"), J = dim0(cb, "c"), nrhs = dim1(Yb, "Y");
  want1(t1b, N, "t1"); want1(t2b, M, "t2"); want1(cb, J, "c"); want2(Ub, N, J, "U"); want2(Vb, M, J, "V");
  want2(Yb, M, nrhs, "Y"); want2(Zb, N, nrhs, "Z"); want3(Fb, M, J, nrhs, "F");
  int rc;
  {
    py::gil_scoped_release nogil;
    rc = LOWER ? c2h_general_matmul_lower(N, M, J, nrhs, cptr(t1b), cptr(t2b), cptr(cb), cptr(Ub), cptr(Vb), cptr(Yb), mptr(Zb), mptr(Fb), 1)
               : c2h_general_matmul_upper(N, M, J, nrhs, cptr(t1b), cptr(t2b), cptr(cb), cptr(Ub), cptr(Vb), cptr(Yb), mptr(Zb), mptr(Fb), 1);
  }
  check(rc);
  return std::make_tuple(Z, F);
}

}  // namespace

PYBIND11_MODULE(backprop, m) {
  m.doc() = "celerite2.backprop drop-in backed by gfx950 HIP kernels (libcelerite2_amd.so)";
  py::register_exception<linalg_exception>(m, "LinAlgError");
  m.def("factor_fwd", &factor_fwd);
  m.def("factor_rev", &factor_rev);
  m.def("solve_lower_fwd", &sweep_fwd<0>);
  m.def("solve_lower_rev", &sweep_rev<0>);
  m.def("solve_upper_fwd", &sweep_fwd<1>);
  m.def("solve_upper_rev", &sweep_rev<1>);
  m.def("matmul_lower_fwd", &sweep_fwd<2>);
  m.def("matmul_lower_rev", &sweep_rev<2>);
  m.def("matmul_upper_fwd", &sweep_fwd<3>);
  m.def("matmul_upper_rev", &sweep_rev<3>);
  m.def("general_matmul_lower_fwd", &general_fwd<true>);
  m.def("general_matmul_upper_fwd", &general_fwd<false>);
  m.attr("__version__") = c2_version();
}
