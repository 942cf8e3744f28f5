/* mrhyde_oracle_expr.c -- TEST INFRASTRUCTURE ONLY: recursive-descent evaluator of deck function strings. */
#include <ctype.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "mrhyde_oracle.h"

typedef struct { const char *p; const double *x, *n; double t, h; int err; } px;

static void ws(px *s) { while (isspace((unsigned char)*s->p)) ++s->p; }
static double p_cmp(px *s);

static double p_primary(px *s) {
  ws(s);
  if (*s->p == '(') {
    ++s->p;
    double v = p_cmp(s);
    ws(s);
    if (*s->p == ')') ++s->p; else s->err = 1;
    return v;
  }
  if (isdigit((unsigned char)*s->p) || *s->p == '.') {
    char *end;
    double v = strtod(s->p, &end);
    if (end == s->p) s->err = 1;
    s->p = end;
    return v;
  }
  if (isalpha((unsigned char)*s->p)) {
    char id[16];
    int k = 0;
    while (isalnum((unsigned char)*s->p) && k < 15) id[k++] = *s->p++;
    id[k] = 0;
    static const char *fn[] = {"sin", "cos", "tan", "exp", "log", "abs", "sqrt", "sinh", "cosh"};
    for (int f = 0; f < 9; ++f)
      if (!strcmp(id, fn[f])) {
        ws(s);
        if (*s->p != '(') { s->err = 1; return 0.0; }
        double a = p_primary(s);
        switch (f) {
          case 0: return sin(a); case 1: return cos(a); case 2: return tan(a); case 3: return exp(a);
          case 4: return log(a); case 5: return fabs(a); case 6: return sqrt(a); case 7: return sinh(a);
          default: return cosh(a);
        }
      }
    if (!strcmp(id, "x")) return s->x[0];
    if (!strcmp(id, "y")) return s->x[1];
    if (!strcmp(id, "z")) return s->x[2];
    if (!strcmp(id, "t")) return s->t;
    if (!strcmp(id, "nx")) return s->n ? s->n[0] : 0.0;
    if (!strcmp(id, "ny")) return s->n ? s->n[1] : 0.0;
    if (!strcmp(id, "nz")) return s->n ? s->n[2] : 0.0;
    if (!strcmp(id, "h")) return s->h;
    if (!strcmp(id, "pi")) return 3.141592653589793238;
  }
  s->err = 1;
  return 0.0;
}

static double p_unary(px *s);
static double p_power(px *s) { /* right-associative, binds tighter than unary minus: -x^2 = -(x^2), 2^-1 allowed */
  double a = p_primary(s);
  ws(s);
  if (*s->p == '^') { ++s->p; return pow(a, p_unary(s)); }
  return a;
}
static double p_unary(px *s) {
  ws(s);
  if (*s->p == '-') { ++s->p; return -p_unary(s); }
  if (*s->p == '+') { ++s->p; return p_unary(s); }
  return p_power(s);
}
static double p_term(px *s) {
  double a = p_unary(s);
  for (;;) {
    ws(s);
    if (*s->p == '*') { ++s->p; a *= p_unary(s); }
    else if (*s->p == '/') { ++s->p; a /= p_unary(s); }
    else return a;
  }
}
static double p_sum(px *s) {
  double a = p_term(s);
  for (;;) {
    ws(s);
    if (*s->p == '+') { ++s->p; a += p_term(s); }
    else if (*s->p == '-') { ++s->p; a -= p_term(s); }
    else return a;
  }
}
static double p_cmp(px *s) {
  double a = p_sum(s);
  for (;;) {
    ws(s);
    if (*s->p == '<' || *s->p == '>') {
      const char c = *s->p++;
      const int eq = (*s->p == '=');
      if (eq) ++s->p;
      const double b = p_sum(s);
      a = (c == '<') ? (eq ? a <= b : a < b) : (eq ? a >= b : a > b);
    } else return a;
  }
}

double orc_eval_expression(const char *expr, const double *xyz, double t, const double *nrm, double h, int *err) {
  double x3[3] = {xyz[0], xyz[1], 0.0};
  px s = {expr, x3, nrm, t, h, 0};
  (void)x3;
  s.x = xyz;
  double v = p_cmp(&s);
  ws(&s);
  if (*s.p) s.err = 1;
  if (err) *err = s.err;
  return s.err ? 0.0 : v;
}
