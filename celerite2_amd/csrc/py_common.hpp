// py_common.hpp -- shared helpers of the `driver` / `backprop` pybind11 modules.
//
// Mirrors the argument handling of the reference bindings
// (python/celerite2/driver.cpp:13-64 and every generated function after it):
//   * every argument is `py::array_t<double, py::array::c_style>` (default
//     forcecast: a non-contiguous / non-f64 argument is silently copied and the
//     result lands in -- and is returned as -- the copy; C-contiguous float64
//     arrays are used in place, which reference tests assert, test_driver.py:13-15);
//   * N = t.shape[0], J = c.shape[0], nrhs = Y.shape[1]; every other argument is
//     checked against them and a mismatch raises std::invalid_argument
//     ("Invalid shape: <name>") -> Python ValueError;
//   * a failed factorisation raises the module's LinAlgError with the message
//     "failed to factorize or solve matrix" (driver.hpp:13-19).
// The numerical work is done by the C-ABI in include/celerite2_amd.h (c2h_*),
// i.e. by the gfx950 HIP kernels; there is no CPU fallback.
#pragma once
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <stdexcept>
#include <string>

#include "../../include/celerite2_amd.h"

namespace py = pybind11;
using Arr = py::array_t<double, py::array::c_style>;

namespace c2py {

// Each module has its OWN exception class (driver.hpp:13-19 defines driver_linalg_exception and
// backprop_linalg_exception separately); distinct C++ types keep pybind11's translators apart.
#ifndef C2PY_LINALG_EXCEPTION
#error "define C2PY_LINALG_EXCEPTION (the module's exception type name) before including py_common.hpp"
#endif
struct C2PY_LINALG_EXCEPTION : public std::exception {
  const char *what() const throw() { return "failed to factorize or solve matrix"; }
};
using linalg_exception = C2PY_LINALG_EXCEPTION;

inline py::ssize_t dim0(const py::buffer_info &b, const char *name) {
  if (b.ndim <= 0) throw std::invalid_argument(std::string("Invalid number of dimensions: ") + name);
  return b.shape[0];
}
inline py::ssize_t dim1(const py::buffer_info &b, const char *name) {
  if (b.ndim <= 1) throw std::invalid_argument(std::string("Invalid number of dimensions: ") + name);
  return b.shape[1];
}
inline void want1(const py::buffer_info &b, py::ssize_t n, const char *name) {
  if (b.ndim != 1 || b.shape[0] != n) throw std::invalid_argument(std::string("Invalid shape: ") + name);
}
inline void want2(const py::buffer_info &b, py::ssize_t n, py::ssize_t m, const char *name) {
  if (b.ndim != 2 || b.shape[0] != n || b.shape[1] != m)
    throw std::invalid_argument(std::string("Invalid shape: ") + name);
}
inline void want3(const py::buffer_info &b, py::ssize_t n, py::ssize_t m, py::ssize_t k, const char *name) {
  if (b.ndim != 3 || b.shape[0] != n || b.shape[1] != m || b.shape[2] != k)
    throw std::invalid_argument(std::string("Invalid shape: ") + name);
}
inline const double *cptr(const py::buffer_info &b) { return (const double *)b.ptr; }
inline double *mptr(const py::buffer_info &b) { return (double *)b.ptr; }

// Translate a C-ABI status into the exception a caller of the reference would see.
inline void check(int rc) {
  if (rc == C2_OK) return;
  if (rc == C2_ERR_INVALID) throw std::invalid_argument("Invalid shape: empty or inconsistent dimensions");
  if (rc == C2_ERR_UNSUPPORTED)
    throw std::invalid_argument("celerite2_amd: J exceeds the supported width (C2_MAX_WIDTH = 128)");
  throw std::runtime_error(std::string("celerite2_amd: HIP error (is an MI355X visible?): ") + c2_last_error());
}

}  // namespace c2py
