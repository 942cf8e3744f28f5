// expression.hpp -- host side of MHA_FUNC_EXPRESSION: compiles a function string of the reference's input decks
// ("8*pi*pi*sin(2*pi*x)*sin(2*pi*y)") into a postfix program the kernels interpret at the integration points.
//
// The reference parses such strings into a DAG of Kokkos views and evaluates one kernel per node
// (FunctionManager::decomposeFunctions / evaluate, src/managers/functionManager.cpp:95-540, 543-760;
// Interpreter::split, src/tools/interpreter.cpp).  Supported here: numbers, the known variables x y z t nx ny nz h pi
// (functionManager.cpp:21), + - * / ^ with the usual precedence (^ binds tightest, right-associative; unary minus),
// < > <= >= (1.0 / 0.0), parentheses and the unary operations sin cos tan exp log abs sqrt sinh cosh (:22).
// Any other identifier goes to the caller's resolver: solution fields of the block -- `e`, `grad(e)[x]`, `div(u)`, `e_t`,
// `u[x]`: the names of Workset::getSolutionField (src/tools/workset.cpp:314-379) -- become EXPR_FIELD / EXPR_FIELD_T
// operands (slot of the point engine's field array), and other named functions of the deck are inlined
// (FunctionManager::decomposeFunctions resolves them into sub-trees the same way, functionManager.cpp:95-540).
// The view reductions max / min / mean / emax / emin / emean are not available: MHA_ERR_INVALID.
#pragma once
#include <functional>
#include <string>
#include <vector>

#include "kernels/device_types.hpp"

namespace mha {

// -> program (ExprOp codes + constant indices) and constants; throws Error(MHA_ERR_INVALID) on anything unsupported
// resolver(identifier, code, consts): append the code that pushes the identifier's value and return true, or return
// false (unknown).  *uses_fields is set when the program contains EXPR_FIELD / EXPR_FIELD_T.
using ExprResolver = std::function<bool(const std::string &, std::vector<int32_t> &, std::vector<double> &)>;
void compile_expression(const std::string &text, std::vector<int32_t> &code, std::vector<double> &consts,
                        const ExprResolver &resolver = ExprResolver(), bool *uses_fields = nullptr);

}  // namespace mha
