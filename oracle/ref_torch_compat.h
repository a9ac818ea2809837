// oracle/ref_torch_compat.h -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Force-included (-include) when oracle/build_ref.py compiles the reference's own,
// unmodified C++ sources from /root/reference/lib/model/csrc.  Those sources call
// AT_DISPATCH_FLOATING_TYPES(tensor.type(), ...) (cpu/ROIAlign_cpu.cpp:242,
// cpu/nms_cpu.cpp:71).  torch >= 2.x dropped the deprecated overload of
// ::detail::scalar_type() that accepted at::DeprecatedTypeProperties, which is the only
// thing that stops these two files compiling against the torch 2.10 headers in this
// image.  This header restores that one overload; it replaces no header, library or
// tool, and changes no arithmetic.
#pragma once
#include <ATen/ATen.h>
#include <ATen/Dispatch.h>
#include <ATen/core/DeprecatedTypeProperties.h>
namespace detail {
inline at::ScalarType scalar_type(const at::DeprecatedTypeProperties& t) {
  return t.scalarType();
}
}  // namespace detail
