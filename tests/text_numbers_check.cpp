// Host check of atom_amd/csrc/text_numbers.h (built and run by tests/test_abi.py): the direct conversion of plain
// decimal tokens against the stream extraction the reference reads its .xyz files with, bit for bit.
// usage: text_numbers_check <scratch directory>
#include "text_numbers.h"
#include <cstdio>
#include <cstring>
#include <random>
#include <sstream>
int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    // random decimal tokens in many formats: the direct conversion against the stream extraction, bit for bit
    std::mt19937_64 rng(5);
    std::string text; std::vector<std::string> tok;
    const char* fmts[] = { "%.1f", "%.4f", "%.10f", "%.15g", "%.17g", "%e", "%.3e", "%g", "%.0f", "%+.6f" };
    for (int n = 0; n < 300000; n++) {
        double x = std::ldexp((double)(rng() >> 11) / 9007199254740992.0 - .5, (int)(rng() % 80) - 40);
        char b[64]; snprintf(b, sizeof b, fmts[rng() % 10], x);
        tok.push_back(b); text += b; text += (n % 3 == 2) ? "\n" : " ";
    }
    for (const char* t : { "0", "-0", "0.0", "-0.000", "5.", ".5", "-.5e3", "1E+2", "007", "1e22", "1e23", "1e-22", "123456789012345", "1234567890123456", "9007199254740993", "4.9e-324" }) { tok.push_back(t); text += t; text += " "; }
    FILE* f = fopen((dir + "/tok.txt").c_str(), "w"); fwrite(text.data(), 1, text.size(), f); fclose(f);
    TextNumbers tn; if (!tn.open((dir + "/tok.txt").c_str()) || !tn.plain()) { printf("not plain\n"); return 1; }
    std::istringstream in(text);
    size_t bad = 0, n = 0;
    for (; n < tok.size(); n++) {
        double a, b = 0; if (!tn.next(a)) { printf("ran out at %zu\n", n); return 1; }
        in >> b; if (in.fail()) { in.clear(); }
        if (memcmp(&a, &b, 8) != 0 && bad++ < 10) printf("MISMATCH %s: %a vs %a\n", tok[n].c_str(), a, b);
    }
    {   // the same file taken whole (several threads, each on a piece of the text): the same values in the same order
        TextNumbers whole; whole.open((dir + "/tok.txt").c_str());
        std::vector<double> v; whole.all(v);
        std::istringstream again(text);
        if (v.size() != tok.size()) { printf("all(): %zu values for %zu tokens\n", v.size(), tok.size()); bad++; }
        for (size_t e = 0; e < v.size() && e < tok.size(); e++) {
            double b = 0; again >> b; if (again.fail()) again.clear();
            if (memcmp(&v[e], &b, 8) != 0 && bad++ < 10) printf("MISMATCH in all() at %zu (%s)\n", e, tok[e].c_str());
        }
    }
    double a; printf("%zu tokens, %zu mismatches, end %d\n", n, bad, (int)tn.next(a));
    { FILE* g = fopen((dir + "/bad.txt").c_str(), "w"); fputs("1 2 1e400", g); fclose(g); std::vector<double> v; bool op;
      if (text_third_of_triples((dir + "/bad.txt").c_str(), 1, v, &op)) { printf("out-of-range token taken\n"); bad++; } }
    for (const char* t : { "1 2 nan", "1 2 3x", "1 2 --3", "1 e5 3", "1 2 0x10", "1 2 1e", "1 2 ." }) {
        FILE* g = fopen((dir + "/bad.txt").c_str(), "w"); fputs(t, g); fclose(g);
        TextNumbers q; q.open((dir + "/bad.txt").c_str()); if (q.plain()) { printf("accepted as plain: %s\n", t); bad++; }
    }
    return bad != 0;
}
