// edgelist2bin -- text edge lists -> the reference's three-file binary graph format.
//
// GraphMiner loads `prefix.meta.txt / prefix.vertex.bin / prefix.edge.bin` (src/common/graph.cc:21-41) and points to a
// converter in another repository for producing them (README.md:104: "mtx, lg, sadj or txt"). This is that tool for the two
// formats the public datasets of the README tables come in:
//   * SNAP text (.txt / .el / .edges): one "u v" pair per line, `#` comment lines, any white space or commas between the ids;
//   * Matrix Market coordinate (.mtx): `%` comment lines, one "rows cols nnz" size line, then 1-based "i j [value]" lines.
// The pairs are symmetrised, self loops dropped, every row sorted ascending and de-duplicated -- the shape the loader asserts
// and every solver assumes ([probe] on inputs/citeseer: rows strictly ascending, symmetric, no self loops) -- and max_degree is
// the longest row: every simple graph with at least one edge satisfies the loader's `max_degree > 0 && max_degree < nv`
// (graph.cc:34), a graph without an edge is refused.
//
//   edgelist2bin [--one-based] [--compact] [--mtx] [--meta-tail "f c e"] <edges.txt> <out prefix>
//   edgelist2bin --dump <prefix> <out.txt>        (the inverse: every undirected edge once, "u v" with u < v, ascending)
//
//   --one-based   ids start at 1 (com-Orkut's SNAP file): 1 is subtracted (implied by --mtx / a .mtx suffix)
//   --compact     renumber the ids that occur densely, order preserved (SNAP files with gaps in the id space); without it
//                 nv = largest id + 1 and unused ids are isolated vertices (no pattern count depends on them)
//   --meta-tail   the last three numbers of meta.txt (feat_len, vertex classes, edge classes); default "0 0 0"
//
// Host-only C++ (no HIP): g++ -O2 -std=c++17.
#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

[[noreturn]] void die(const std::string &msg) {
  std::fprintf(stderr, "edgelist2bin: %s\n", msg.c_str());
  std::exit(1);
}

std::vector<char> slurp(const std::string &path) {
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f) die("cannot open " + path);
  std::vector<char> buf;
  char tmp[1 << 16];
  size_t n;
  while ((n = std::fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
  std::fclose(f);
  buf.push_back('\n');
  return buf;
}

void write_file(const std::string &path, const void *p, size_t bytes) {
  FILE *f = std::fopen(path.c_str(), "wb");
  if (!f) die("cannot write " + path);
  if (bytes && std::fwrite(p, 1, bytes, f) != bytes) die("short write to " + path);
  std::fclose(f);
}

// the unsigned integers of one text line (at most `cap`); returns how many were found, -1 on a malformed token
int line_ints(const char *b, const char *e, uint64_t *out, int cap) {
  int n = 0;
  while (b < e) {
    while (b < e && (std::isspace((unsigned char)*b) || *b == ',' || *b == ';')) ++b;
    if (b >= e) break;
    if (!std::isdigit((unsigned char)*b)) {
      if (n >= 2) break;  // a trailing weight / label (mtx values, "1.0", "-3"): ignored
      return -1;
    }
    uint64_t v = 0;
    while (b < e && std::isdigit((unsigned char)*b)) v = v * 10 + (uint64_t)(*b++ - '0');
    if (b < e && (*b == '.' || *b == 'e' || *b == 'E')) {  // a real number: only acceptable as a trailing value
      if (n >= 2) break;
      return -1;
    }
    if (n < cap) out[n] = v;
    ++n;
  }
  return n;
}

int dump(const std::string &prefix, const std::string &out_path) {
  FILE *fm = std::fopen((prefix + ".meta.txt").c_str(), "r");
  if (!fm) die("cannot open " + prefix + ".meta.txt");
  long long nv = 0, ne = 0;
  if (std::fscanf(fm, "%lld %lld", &nv, &ne) != 2) die("bad meta.txt");
  std::fclose(fm);
  std::vector<char> vb = slurp(prefix + ".vertex.bin"), eb = slurp(prefix + ".edge.bin");
  if (vb.size() - 1 != (size_t)(nv + 1) * 8 || eb.size() - 1 != (size_t)ne * 4) die("vertex.bin / edge.bin size does not match meta.txt");
  const int64_t *rp = reinterpret_cast<const int64_t *>(vb.data());
  const int32_t *ci = reinterpret_cast<const int32_t *>(eb.data());
  FILE *fo = std::fopen(out_path.c_str(), "w");
  if (!fo) die("cannot write " + out_path);
  std::fprintf(fo, "# Undirected graph: %s\n# Nodes: %lld Edges: %lld\n", prefix.c_str(), nv, ne / 2);
  for (long long u = 0; u < nv; ++u)
    for (int64_t e = rp[u]; e < rp[u + 1]; ++e)
      if (ci[e] > u) std::fprintf(fo, "%lld\t%d\n", u, ci[e]);
  std::fclose(fo);
  return 0;
}

}  // namespace

int main(int argc, char **argv) {
  bool one_based = false, compact = false, mtx = false, do_dump = false;
  std::string meta_tail = "0 0 0";
  std::vector<std::string> pos;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--one-based") one_based = true;
    else if (a == "--compact") compact = true;
    else if (a == "--mtx") mtx = true;
    else if (a == "--dump") do_dump = true;
    else if (a == "--meta-tail" && i + 1 < argc) meta_tail = argv[++i];
    else if (a.rfind("--", 0) == 0) die("unknown option " + a);
    else pos.push_back(a);
  }
  if (pos.size() != 2) {
    std::fprintf(stderr, "usage: %s [--one-based] [--compact] [--mtx] [--meta-tail \"f c e\"] <edges.txt> <out prefix>\n"
                         "       %s --dump <prefix> <out.txt>\n", argv[0], argv[0]);
    return 2;
  }
  if (do_dump) return dump(pos[0], pos[1]);
  const std::string in = pos[0], prefix = pos[1];
  if (in.size() > 4 && in.compare(in.size() - 4, 4, ".mtx") == 0) mtx = true;
  if (mtx) one_based = true;

  const std::vector<char> buf = slurp(in);
  std::vector<uint64_t> keys;  // (u << 32) | v, both directions
  keys.reserve(buf.size() / 6);
  bool size_line_pending = mtx;
  uint64_t max_id = 0;
  long long lineno = 0, self_loops = 0;
  for (const char *p = buf.data(), *end = buf.data() + buf.size(); p < end;) {
    const char *nl = static_cast<const char *>(std::memchr(p, '\n', (size_t)(end - p)));
    const char *le = nl ? nl : end;
    ++lineno;
    const char *q = p;
    while (q < le && std::isspace((unsigned char)*q)) ++q;
    if (q < le && *q != '#' && *q != '%') {
      uint64_t v[3];
      const int n = line_ints(q, le, v, 3);
      if (n < 2) die(in + ":" + std::to_string(lineno) + ": expected two vertex ids");
      if (size_line_pending) {
        size_line_pending = false;  // "rows cols nnz"
      } else {
        uint64_t a = v[0], b = v[1];
        if (one_based) {
          if (a == 0 || b == 0) die(in + ":" + std::to_string(lineno) + ": id 0 in a 1-based file");
          --a;
          --b;
        }
        if (a >= 0x7fffffffull || b >= 0x7fffffffull) die(in + ":" + std::to_string(lineno) + ": vertex id does not fit int32");
        if (a == b) {
          ++self_loops;
        } else {
          keys.push_back((a << 32) | b);
          keys.push_back((b << 32) | a);
        }
        // (without --compact the id space is [0, largest id]: a vertex that only occurs in a self loop is still a vertex)
        max_id = std::max(max_id, std::max(a, b));
      }
    }
    p = le + 1;
  }
  std::sort(keys.begin(), keys.end());
  const size_t before = keys.size();
  keys.erase(std::unique(keys.begin(), keys.end()), keys.end());

  uint64_t nv = keys.empty() && self_loops == 0 ? 0 : max_id + 1;
  if (compact) {  // dense renumbering of the ids that have an edge, order preserved
    std::vector<uint32_t> ids;
    ids.reserve(keys.size() / 8 + 1);
    for (uint64_t k : keys) {  // sources in ascending order: every endpoint is the source of one of the two directions
      const uint32_t u = (uint32_t)(k >> 32);
      if (ids.empty() || ids.back() != u) ids.push_back(u);
    }
    for (uint64_t &k : keys) {
      const uint32_t u = (uint32_t)(k >> 32), v = (uint32_t)k;
      const uint64_t nu = (uint64_t)(std::lower_bound(ids.begin(), ids.end(), u) - ids.begin());
      const uint64_t nw = (uint64_t)(std::lower_bound(ids.begin(), ids.end(), v) - ids.begin());
      k = (nu << 32) | nw;
    }
    nv = ids.size();  // (the map is monotone: the keys stay sorted)
  }
  if (nv >= 0x7fffffffull) die("too many vertices for int32 ids");

  std::vector<int64_t> rp((size_t)nv + 1, 0);
  std::vector<int32_t> ci(keys.size());
  for (size_t i = 0; i < keys.size(); ++i) {
    ++rp[(size_t)(keys[i] >> 32) + 1];
    ci[i] = (int32_t)(uint32_t)keys[i];
  }
  int64_t max_deg = 0;
  for (size_t v = 0; v < (size_t)nv; ++v) {
    max_deg = std::max(max_deg, rp[v + 1]);
    rp[v + 1] += rp[v];
  }
  if (!(max_deg > 0 && (uint64_t)max_deg < nv)) die("the graph has no edge: the loader needs 0 < max_degree < nv (src/common/graph.cc:34)");

  // meta.txt: nv, ne, sizeof(vid) sizeof(eid) sizeof(vlabel) sizeof(elabel), max_degree, feat_len, #vertex classes, #edge classes
  // (src/common/graph.cc:27-29; layout of inputs/citeseer/graph.meta.txt)
  {
    int t[3] = {0, 0, 0};
    if (std::sscanf(meta_tail.c_str(), "%d %d %d", &t[0], &t[1], &t[2]) != 3) die("--meta-tail wants three integers");
    char m[256];
    const int n = std::snprintf(m, sizeof m, "%llu\n%zu\n4 8 1 2\n%lld\n%d\n%d\n%d\n", (unsigned long long)nv, keys.size(), (long long)max_deg, t[0], t[1], t[2]);
    write_file(prefix + ".meta.txt", m, (size_t)n);
  }
  write_file(prefix + ".vertex.bin", rp.data(), rp.size() * sizeof(int64_t));
  write_file(prefix + ".edge.bin", ci.data(), ci.size() * sizeof(int32_t));
  std::printf("|V|: %llu, |E|: %zu, Max Degree: %lld\n", (unsigned long long)nv, keys.size(), (long long)max_deg);  // Graph::print_meta_data
  std::printf("%lld input lines, %zu directed pairs kept (%zu duplicates and %lld self loops dropped)\n", lineno - 1, keys.size(), before - keys.size(), self_loops);
  return 0;
}
