// py_driver.cpp -- `celerite2_amd.driver`: drop-in for the reference's
// `celerite2.driver` pybind11 module (python/celerite2/driver.cpp:13-499).
// Same function names, positional argument order, shape checks, in-place
// semantics, return values and LinAlgError; the work runs on the GPU through
// the C-ABI (include/celerite2_amd.h).  See INTEGRATION.md.
#define C2PY_LINALG_EXCEPTION driver_linalg_exception
#include "py_common.hpp"

using namespace c2py;

namespace {

// driver.factor -- driver.cpp:13-64
auto factor(Arr t, Arr c, Arr a, Arr U, Arr V, Arr d, Arr W) {
  py::buffer_info tb = t.request(), cb = c.request(), ab = a.request(), Ub = U.request(), Vb = V.request(),
                  db = d.request(), Wb = W.request();
  const py::ssize_t N = dim0(tb, "t"), J = dim0(cb, "c");
  want1(tb, N, "t"); want1(cb, J, "c"); want1(ab, N, "a");
  want2(Ub, N, J, "U"); want2(Vb, N, J, "V"); want1(db, N, "d"); want2(Wb, N, J, "W");
  int64_t flag = 0;
  int rc;
  {
    py::gil_scoped_release nogil;
    rc = c2h_factor(N, J, cptr(tb), cptr(cb), cptr(ab), cptr(Ub), cptr(Vb), mptr(db), mptr(Wb), nullptr, &flag);
  }
  check(rc);
  if (flag) throw linalg_exception();
  return std::make_tuple(d, W);
}

// driver.solve_lower / solve_upper / matmul_lower / matmul_upper -- driver.cpp:66-292
template <int OP>
auto sweep(Arr t, Arr c, Arr U, Arr W, Arr Y, Arr Z) {
  py::buffer_info tb = t.request(), cb = c.request(), Ub = U.request(), Wb = W.request(), Yb = Y.request(),
                  Zb = Z.request();
  const py::ssize_t N = dim0(tb, "t"), J = dim0(cb, "c"), nrhs = dim1(Yb, "Y");
  want1(tb, N, "t"); want1(cb, J, "c"); want2(Ub, N, J, "U"); want2(Wb, N, J, OP < 2 ? "W" : "V");
  want2(Yb, N, nrhs, "Y"); want2(Zb, N, nrhs, "Z");
  int rc;
  {
    py::gil_scoped_release nogil;
    if (OP == 0) rc = c2h_solve_lower(N, J, nrhs, cptr(tb), cptr(cb), cptr(Ub), cptr(Wb), cptr(Yb), mptr(Zb), nullptr);
    else if (OP == 1) rc = c2h_solve_upper(N, J, nrhs, cptr(tb), cptr(cb), cptr(Ub), cptr(Wb), cptr(Yb), mptr(Zb), nullptr);
    else if (OP == 2) rc = c2h_matmul_lower(N, J, nrhs, cptr(tb), cptr(cb), cptr(Ub), cptr(Wb), cptr(Yb), mptr(Zb), nullptr, 0);
    else rc = c2h_matmul_upper(N, J, nrhs, cptr(tb), cptr(cb), cptr(Ub), cptr(Wb), cptr(Yb), mptr(Zb), nullptr, 0);
  }
  check(rc);
  return Z;
}

// driver.general_matmul_lower / upper -- driver.cpp:294-420
template <bool LOWER>
auto general(Arr t1, Arr t2, Arr c, Arr U, Arr V, Arr Y, Arr Z) {
  py::buffer_info t1b = t1.request(), t2b = t2.request(), cb = c.request(), Ub = U.request(), Vb = V.request(),
                  Yb = Y.request(), Zb = Z.request();
  const py::ssize_t N = dim0(t1b, "t1"), M = dim0(t2b, "t2"), J = dim0(cb, "c"), nrhs = dim1(Yb, "Y");
  want1(t1b, N, "t1"); want1(t2b, M, "t2"); want1(cb, J, "c"); want2(Ub, N, J, "U"); want2(Vb, M, J, "V");
  want2(Yb, M, nrhs, "Y"); want2(Zb, N, nrhs, "Z");
  int rc;
  {
    py::gil_scoped_release nogil;
    rc = LOWER ? c2h_general_matmul_lower(N, M, J, nrhs, cptr(t1b), cptr(t2b), cptr(cb), cptr(Ub), cptr(Vb), cptr(Yb), mptr(Zb), nullptr, 0)
               : c2h_general_matmul_upper(N, M, J, nrhs, cptr(t1b), cptr(t2b), cptr(cb), cptr(Ub), cptr(Vb), cptr(Yb), mptr(Zb), nullptr, 0);
  }
  check(rc);
  return Z;
}

// driver.get_celerite_matrices -- driver.cpp:422-477
auto get_celerite_matrices(Arr ar, Arr ac, Arr bc, Arr dc, Arr x, Arr diag, Arr a, Arr U, Arr V) {
  py::buffer_info arb = ar.request(), acb = ac.request(), bcb = bc.request(), dcb = dc.request(), xb = x.request(),
                  diagb = diag.request(), ab = a.request(), Ub = U.request(), Vb = V.request();
  if (arb.ndim != 1 || acb.ndim != 1 || bcb.ndim != 1 || dcb.ndim != 1 || xb.ndim != 1 || diagb.ndim != 1 ||
      ab.ndim != 1 || Ub.ndim != 2 || Vb.ndim != 2)
    throw std::invalid_argument("dimension mismatch: wrong number of dimensions");
  const py::ssize_t N = xb.shape[0], Jr = arb.shape[0], Jc = acb.shape[0], J = Jr + 2 * Jc;
  if (bcb.shape[0] != Jc) throw std::invalid_argument("dimension mismatch: bc");
  if (dcb.shape[0] != Jc) throw std::invalid_argument("dimension mismatch: dc");
  if (diagb.shape[0] != N) throw std::invalid_argument("dimension mismatch: diag");
  if (ab.shape[0] != N) throw std::invalid_argument("dimension mismatch: a");
  if (Ub.shape[0] != N || Ub.shape[1] != J) throw std::invalid_argument("dimension mismatch: U");
  if (Vb.shape[0] != N || Vb.shape[1] != J) throw std::invalid_argument("dimension mismatch: V");
  int rc;
  {
    py::gil_scoped_release nogil;
    rc = c2h_get_celerite_matrices(N, Jr, Jc, cptr(arb), cptr(acb), cptr(bcb), cptr(dcb), cptr(xb), cptr(diagb),
                                   mptr(ab), mptr(Ub), mptr(Vb));
  }
  check(rc);
  return std::make_tuple(a, U, V);
}

}  // namespace

PYBIND11_MODULE(driver, m) {
  m.doc() = "celerite2.driver drop-in backed by gfx950 HIP kernels (libcelerite2_amd.so)";
  py::register_exception<linalg_exception>(m, "LinAlgError");
  m.def("factor", &factor);
  m.def("solve_lower", &sweep<0>);
  m.def("solve_upper", &sweep<1>);
  m.def("matmul_lower", &sweep<2>);
  m.def("matmul_upper", &sweep<3>);
  m.def("general_matmul_lower", &general<true>);
  m.def("general_matmul_upper", &general<false>);
  m.def("get_celerite_matrices", &get_celerite_matrices);
  m.attr("__version__") = c2_version();
}
