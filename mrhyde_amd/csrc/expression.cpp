#include "expression.hpp"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdlib>

#include "common.hpp"

namespace mha {
namespace {

struct Tok {
  enum Kind { NUM, VAR, FUNC, OP, LP, RP, SUB } kind;
  double num = 0.0;
  int code = 0;  // ExprOp for VAR / FUNC / OP
  std::vector<int32_t> sub_code;  // SUB: code that pushes the value of a resolved identifier (field, named function)
  std::vector<double> sub_consts;
};

int prec(int op) {
  switch (op) {
    case EXPR_LT: case EXPR_GT: case EXPR_LE: case EXPR_GE: return 1;
    case EXPR_ADD: case EXPR_SUB: return 2;
    case EXPR_MUL: case EXPR_DIV: return 3;
    case EXPR_NEG: return 4;
    case EXPR_POW: return 5;
    default: return 0;
  }
}
bool right_assoc(int op) { return op == EXPR_POW || op == EXPR_NEG; }

}  // namespace

void compile_expression(const std::string &text, std::vector<int32_t> &code, std::vector<double> &consts,
                        const ExprResolver &resolver, bool *uses_fields) {
  static const struct { const char *name; int code; bool func; } names[] = {
      {"x", EXPR_X, false},   {"y", EXPR_Y, false},   {"z", EXPR_Z, false},   {"t", EXPR_T, false},
      {"nx", EXPR_NX, false}, {"ny", EXPR_NY, false}, {"nz", EXPR_NZ, false}, {"h", EXPR_H, false},
      {"pi", EXPR_PI, false}, {"sin", EXPR_SIN, true}, {"cos", EXPR_COS, true}, {"tan", EXPR_TAN, true},
      {"exp", EXPR_EXP, true}, {"log", EXPR_LOG, true}, {"abs", EXPR_ABS, true}, {"sqrt", EXPR_SQRT, true},
      {"sinh", EXPR_SINH, true}, {"cosh", EXPR_COSH, true}};
  // ---- tokens ----
  std::vector<Tok> toks;
  size_t i = 0;
  while (i < text.size()) {
    const char c = text[i];
    if (std::isspace(static_cast<unsigned char>(c))) { ++i; continue; }
    if (std::isdigit(static_cast<unsigned char>(c)) || c == '.') {
      char *end = nullptr;
      Tok t;
      t.kind = Tok::NUM;
      t.num = std::strtod(text.c_str() + i, &end);
      MHA_REQUIRE(end != text.c_str() + i, MHA_ERR_INVALID, "bad number in expression '" << text << "'");
      i = static_cast<size_t>(end - text.c_str());
      toks.push_back(t);
      continue;
    }
    if (std::isalpha(static_cast<unsigned char>(c)) || c == '_') {
      size_t j = i;
      while (j < text.size() && (std::isalnum(static_cast<unsigned char>(text[j])) || text[j] == '_' || text[j] == '[' || text[j] == ']')) ++j;
      std::string id = text.substr(i, j - i);
      // field names with an argument: grad(e)[x], div(u), curl(B)[z] -- one identifier up to the closing parenthesis
      // and an optional component
      if ((id == "grad" || id == "div" || id == "curl") && j < text.size() && text[j] == '(') {
        const size_t close = text.find(')', j);
        MHA_REQUIRE(close != std::string::npos, MHA_ERR_INVALID, "unbalanced '(' after '" << id << "' in expression '" << text << "'");
        j = close + 1;
        if (j < text.size() && text[j] == '[') {
          const size_t cb = text.find(']', j);
          MHA_REQUIRE(cb != std::string::npos, MHA_ERR_INVALID, "unbalanced '[' in expression '" << text << "'");
          j = cb + 1;
        }
        id = text.substr(i, j - i);
        id.erase(std::remove_if(id.begin(), id.end(), [](char ch) { return std::isspace(static_cast<unsigned char>(ch)); }), id.end());
      }
      Tok t;
      bool found = false;
      for (const auto &nm : names)
        if (id == nm.name) { t.kind = nm.func ? Tok::FUNC : Tok::VAR; t.code = nm.code; found = true; }
      if (!found && resolver) {
        t.kind = Tok::SUB;
        found = resolver(id, t.sub_code, t.sub_consts);
      }
      MHA_REQUIRE(found, MHA_ERR_INVALID,
                  "expression '" << text << "': '" << id << "' is not available (known: x y z t nx ny nz h pi, sin cos tan "
                                                            "exp log abs sqrt sinh cosh, the block's solution fields and the "
                                                            "deck's other functions; no view reductions)");
      toks.push_back(t);
      i = j;
      continue;
    }
    Tok t;
    t.kind = Tok::OP;
    if (c == '(') t.kind = Tok::LP;
    else if (c == ')') t.kind = Tok::RP;
    else if (c == '+') t.code = EXPR_ADD;
    else if (c == '-') t.code = EXPR_SUB;
    else if (c == '*') t.code = EXPR_MUL;
    else if (c == '/') t.code = EXPR_DIV;
    else if (c == '^') t.code = EXPR_POW;
    else if (c == '<' || c == '>') {
      const bool eq = i + 1 < text.size() && text[i + 1] == '=';
      t.code = c == '<' ? (eq ? EXPR_LE : EXPR_LT) : (eq ? EXPR_GE : EXPR_GT);
      if (eq) ++i;
    } else {
      MHA_REQUIRE(false, MHA_ERR_INVALID, "unexpected character '" << c << "' in expression '" << text << "'");
    }
    toks.push_back(t);
    ++i;
  }
  // ---- shunting yard ----
  code.clear();
  consts.clear();
  std::vector<Tok> stack;
  bool expect_operand = true;
  auto emit = [&](const Tok &t) {
    if (t.kind == Tok::NUM) { code.push_back(EXPR_CONST); code.push_back(static_cast<int32_t>(consts.size())); consts.push_back(t.num); }
    else if (t.kind == Tok::SUB) {  // inline: constant indices move behind this program's
      const int32_t shift = static_cast<int32_t>(consts.size());
      for (size_t k = 0; k < t.sub_code.size(); ++k) {
        const int32_t op = t.sub_code[k];
        if (op == EXPR_END) break;
        code.push_back(op);
        if (op == EXPR_CONST) code.push_back(t.sub_code[++k] + shift);
        else if (op == EXPR_FIELD || op == EXPR_FIELD_T) code.push_back(t.sub_code[++k]);
      }
      consts.insert(consts.end(), t.sub_consts.begin(), t.sub_consts.end());
    }
    else code.push_back(t.code);
  };
  for (Tok t : toks) {
    switch (t.kind) {
      case Tok::NUM: case Tok::VAR: case Tok::SUB:
        MHA_REQUIRE(expect_operand, MHA_ERR_INVALID, "missing operator in expression '" << text << "'");
        emit(t);
        expect_operand = false;
        break;
      case Tok::FUNC:
        MHA_REQUIRE(expect_operand, MHA_ERR_INVALID, "missing operator in expression '" << text << "'");
        stack.push_back(t);
        break;
      case Tok::LP:
        MHA_REQUIRE(expect_operand, MHA_ERR_INVALID, "missing operator in expression '" << text << "'");
        stack.push_back(t);
        break;
      case Tok::RP:
        MHA_REQUIRE(!expect_operand, MHA_ERR_INVALID, "empty parentheses in expression '" << text << "'");
        while (!stack.empty() && stack.back().kind != Tok::LP) { emit(stack.back()); stack.pop_back(); }
        MHA_REQUIRE(!stack.empty(), MHA_ERR_INVALID, "unbalanced ')' in expression '" << text << "'");
        stack.pop_back();
        if (!stack.empty() && stack.back().kind == Tok::FUNC) { emit(stack.back()); stack.pop_back(); }
        break;
      case Tok::OP:
        if (expect_operand) {  // unary sign: a prefix operator is pushed as it is, nothing on the stack can apply yet
          MHA_REQUIRE(t.code == EXPR_SUB || t.code == EXPR_ADD, MHA_ERR_INVALID, "misplaced operator in expression '" << text << "'");
          if (t.code == EXPR_SUB) { t.code = EXPR_NEG; stack.push_back(t); }
          break;
        }
        while (!stack.empty() && stack.back().kind == Tok::OP &&
               (prec(stack.back().code) > prec(t.code) || (prec(stack.back().code) == prec(t.code) && !right_assoc(t.code)))) {
          emit(stack.back());
          stack.pop_back();
        }
        stack.push_back(t);
        expect_operand = true;
        break;
    }
  }
  MHA_REQUIRE(!expect_operand, MHA_ERR_INVALID, "expression '" << text << "' ends with an operator");
  while (!stack.empty()) {
    MHA_REQUIRE(stack.back().kind == Tok::OP, MHA_ERR_INVALID, "unbalanced '(' in expression '" << text << "'");
    emit(stack.back());
    stack.pop_back();
  }
  code.push_back(EXPR_END);
  // ---- dry run: stack depth ----
  int depth = 0, maxd = 0;
  bool fields = false;
  for (size_t k = 0; k < code.size(); ++k) {
    const int op = code[k];
    if (op == EXPR_END) break;
    if (op == EXPR_CONST || op == EXPR_FIELD || op == EXPR_FIELD_T) { ++k; ++depth; if (op != EXPR_CONST) fields = true; }
    else if (op >= EXPR_X && op <= EXPR_PI) ++depth;
    else if (op >= EXPR_ADD && op <= EXPR_GE) --depth;
    maxd = depth > maxd ? depth : maxd;
  }
  MHA_REQUIRE(depth == 1 && maxd <= kExprStack, MHA_ERR_INVALID,
              "expression '" << text << "' is malformed or needs more than " << kExprStack << " stack entries");
  if (uses_fields) *uses_fields = fields;
}

}  // namespace mha
