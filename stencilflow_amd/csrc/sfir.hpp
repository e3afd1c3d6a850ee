// sfir.hpp — in-memory form and parser of SFIR, the program record the Python
// front end hands to the backend (grammar: stencilflow_amd/lowering.py).
// It carries what the reference passes to DaCe per operator through
// `_generate_stencil` (reference stencilflow/sdfg_generator.py:68-176): accesses
// with relative offsets, boundary conditions, the statements, the output type.
#pragma once

#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace sf {

struct Error : std::runtime_error {
  int status;
  Error(int st, const std::string& msg) : std::runtime_error(msg), status(st) {}
};

enum class DT { F32, F64, I32, I64 };

inline DT parse_dt(const std::string& s) {
  if (s == "f32") return DT::F32;
  if (s == "f64") return DT::F64;
  if (s == "i32") return DT::I32;
  if (s == "i64") return DT::I64;
  throw Error(-1, "SFIR: unknown data type '" + s + "'");
}
inline const char* ctype_of(DT d) {
  switch (d) {
    case DT::F32: return "float";
    case DT::F64: return "double";
    case DT::I32: return "int";
    default: return "long long";
  }
}
inline const char* short_of(DT d) {
  switch (d) {
    case DT::F32: return "f32";
    case DT::F64: return "f64";
    case DT::I32: return "i32";
    default: return "i64";
  }
}
inline size_t size_of(DT d) { return (d == DT::F32 || d == DT::I32) ? 4 : 8; }

struct Scalar {
  std::string name;
  DT dt;
  bool is_const = false;
  std::string literal;  // C literal when is_const
  int input_index = -1; // position among run-time scalars
};

enum class Role { Input, Temp, Output };

struct Field {
  std::string name;
  DT dt;
  bool has[3] = {false, false, false};  // over the normalised (i,j,k) dims
  Role role = Role::Temp;
  int io_index = -1;  // position among array inputs / outputs
  bool full() const { return has[0] && has[1] && has[2]; }
};

struct Access {
  std::string var, field;
  DT vtype;
  std::string bckind, bcval;
  int off[3] = {0, 0, 0};
  bool centre() const { return off[0] == 0 && off[1] == 0 && off[2] == 0; }
};

struct Let {
  std::string ctype, var, expr;
};

struct Kernel {
  std::string name;
  DT dt;
  std::vector<Access> acc;
  std::vector<std::string> uses;
  std::vector<Let> lets;
  std::string ret;
};

struct Program {
  std::string name;
  int nd = 3;
  // Normalised to 3 internal dims (I0 = slab/stream axis, I1 = rows, I2 =
  // contiguous columns): 3-D (i,j,k) -> (0,1,2); 2-D (j,k) -> (0,2) with
  // n[1] = 1; 1-D (k) -> (2).
  long long n[3] = {1, 1, 1};
  int dim_of(int own) const {
    static const int map3[3] = {0, 1, 2}, map2[2] = {0, 2}, map1[1] = {2};
    return nd == 3 ? map3[own] : nd == 2 ? map2[own] : map1[own];
  }
  std::vector<Scalar> scalars;
  std::vector<Field> fields;
  std::vector<Kernel> kernels;
  std::map<std::string, int> field_ix, scalar_ix;
  int num_inputs = 0, num_outputs = 0, num_scalar_inputs = 0;

  const Field& field(const std::string& nm) const {
    auto it = field_ix.find(nm);
    if (it == field_ix.end()) throw Error(-1, "SFIR: unknown field '" + nm + "'");
    return fields[it->second];
  }
};

inline std::vector<std::string> split_ws(const std::string& line) {
  std::istringstream is(line);
  std::vector<std::string> out;
  std::string tok;
  while (is >> tok) out.push_back(tok);
  return out;
}

inline Program parse_sfir(const std::string& text) {
  Program P;
  std::istringstream is(text);
  std::string line;
  Kernel* cur = nullptr;
  bool header = false;
  int lineno = 0;
  auto bad = [&](const std::string& why) {
    return Error(-1, "SFIR line " + std::to_string(lineno) + ": " + why);
  };
  while (std::getline(is, line)) {
    ++lineno;
    auto t = split_ws(line);
    if (t.empty()) continue;
    const std::string& kw = t[0];
    if (kw == "sfir") {
      if (t.size() != 2 || t[1] != "1") throw bad("unsupported SFIR version");
      header = true;
    } else if (!header) {
      throw bad("missing 'sfir 1' header");
    } else if (kw == "program") {
      if (t.size() != 2) throw bad("program <name>");
      P.name = t[1];
    } else if (kw == "dims") {
      if (t.size() < 3) throw bad("dims <nd> <n...>");
      P.nd = std::stoi(t[1]);
      if (P.nd < 1 || P.nd > 3 || (int)t.size() != 2 + P.nd) throw bad("bad dims record");
      for (int d = 0; d < P.nd; ++d) {
        long long v = std::stoll(t[2 + d]);
        if (v < 1) throw bad("dimension must be positive");
        P.n[P.dim_of(d)] = v;
      }
    } else if (kw == "scalar") {
      if (t.size() < 4) throw bad("scalar <name> <dtype> input|const <lit>");
      Scalar s;
      s.name = t[1];
      s.dt = parse_dt(t[2]);
      if (t[3] == "const") {
        if (t.size() != 5) throw bad("scalar const needs a literal");
        s.is_const = true;
        s.literal = t[4];
      } else if (t[3] == "input") {
        s.input_index = P.num_scalar_inputs++;
      } else {
        throw bad("scalar kind must be input or const");
      }
      P.scalar_ix[s.name] = (int)P.scalars.size();
      P.scalars.push_back(s);
    } else if (kw == "field") {
      if (t.size() != 5) throw bad("field <name> <dtype> <mask> <role>");
      Field f;
      f.name = t[1];
      f.dt = parse_dt(t[2]);
      if ((int)t[3].size() != P.nd) throw bad("field mask length != nd");
      for (int d = 0; d < 3; ++d) f.has[d] = true;  // padded dims: extent 1
      for (int d = 0; d < P.nd; ++d) f.has[P.dim_of(d)] = (t[3][d] == '1');
      if (t[4] == "input") {
        f.role = Role::Input;
        f.io_index = P.num_inputs++;
      } else if (t[4] == "output") {
        f.role = Role::Output;
        f.io_index = P.num_outputs++;
      } else if (t[4] == "temp") {
        f.role = Role::Temp;
      } else {
        throw bad("field role must be input, temp or output");
      }
      if (P.field_ix.count(f.name)) throw bad("duplicate field " + f.name);
      P.field_ix[f.name] = (int)P.fields.size();
      P.fields.push_back(f);
    } else if (kw == "kernel") {
      if (t.size() != 3) throw bad("kernel <name> <dtype>");
      if (cur) throw bad("kernel inside kernel");
      P.kernels.emplace_back();
      cur = &P.kernels.back();
      cur->name = t[1];
      cur->dt = parse_dt(t[2]);
    } else if (kw == "acc") {
      if (!cur) throw bad("acc outside kernel");
      if ((int)t.size() != 6 + P.nd) throw bad("acc record has wrong arity");
      Access a;
      a.var = t[1];
      a.field = t[2];
      a.vtype = parse_dt(t[3]);
      a.bckind = t[4];
      a.bcval = t[5];
      const Field& f = P.field(a.field);
      for (int d = 0; d < P.nd; ++d) {
        const std::string& o = t[6 + d];
        const int dim = P.dim_of(d);
        if (o == "x") {
          if (f.has[dim]) throw bad("offset 'x' for a dimension the field has");
          a.off[dim] = 0;
        } else {
          if (!f.has[dim]) throw bad("offset given for a dimension the field lacks");
          a.off[dim] = std::stoi(o);
        }
      }
      if (a.bckind != "none" && a.bckind != "constant" && a.bckind != "shrink" &&
          a.bckind != "copy")
        throw Error(-2, "Unsupported boundary condition type: " + a.bckind);
      cur->acc.push_back(a);
    } else if (kw == "use") {
      if (!cur || t.size() != 2) throw bad("use <scalar>");
      if (!P.scalar_ix.count(t[1])) throw bad("unknown scalar " + t[1]);
      cur->uses.push_back(t[1]);
    } else if (kw == "let") {
      if (!cur) throw bad("let outside kernel");
      // let <ctype...> <var> = <expr>
      size_t eq = line.find(" = ");
      if (eq == std::string::npos) throw bad("let without ' = '");
      auto lhs = split_ws(line.substr(0, eq));
      if (lhs.size() < 3) throw bad("let <ctype> <var> = <expr>");
      Let l;
      l.var = lhs.back();
      for (size_t i = 1; i + 1 < lhs.size(); ++i) l.ctype += (i > 1 ? " " : "") + lhs[i];
      l.expr = line.substr(eq + 3);
      cur->lets.push_back(l);
    } else if (kw == "ret") {
      if (!cur || t.size() != 2) throw bad("ret <var>");
      cur->ret = t[1];
    } else if (kw == "end") {
      if (!cur) throw bad("end outside kernel");
      if (cur->ret.empty()) throw bad("kernel without ret");
      if (!P.field_ix.count(cur->name)) throw bad("kernel without a field record");
      cur = nullptr;
    } else {
      throw bad("unknown record '" + kw + "'");
    }
  }
  if (cur) throw Error(-1, "SFIR: unterminated kernel " + cur->name);
  if (!header) throw Error(-1, "SFIR: empty program");
  if (P.kernels.empty()) throw Error(-1, "SFIR: program has no kernels");
  return P;
}

}  // namespace sf
