// expression.hpp -- host side of MHA_FUNC_EXPRESSION: compiles a function string of the reference's input decks
// ("8*pi*pi*sin(2*pi*x)*sin(2*pi*y)") into a postfix program the kernels interpret at the integration points.
//
// The reference parses such strings into a DAG of Kokkos views and evaluates one kernel per node
// (FunctionManager::decomposeFunctions / evaluate, src/managers/functionManager.cpp:95-540, 543-760;
// Interpreter::split, src/tools/interpreter.cpp).  Supported here: numbers, the known variables x y z t nx ny nz h pi
// (functionManager.cpp:21), + - * / ^ with the usual precedence (^ binds tightest, right-associative; unary minus),
// < > <= >= (1.0 / 0.0), parentheses and the unary operations sin cos tan exp log abs sqrt sinh cosh (:22).
// Field-dependent terms (solution variables, other named functions) and the view reductions max / min / mean /
// emax / emin / emean are not available: the string is rejected with MHA_ERR_INVALID.
#pragma once
#include <string>
#include <vector>

#include "kernels/device_types.hpp"

namespace mha {

// -> program (ExprOp codes + constant indices) and constants; throws Error(MHA_ERR_INVALID) on anything unsupported
void compile_expression(const std::string &text, std::vector<int32_t> &code, std::vector<double> &consts);

}  // namespace mha
