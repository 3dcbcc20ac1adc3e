// Numbers of a whitespace-separated text file (the .xyz inputs of the models), read the way `stream >> double`
// reads them, but an order of magnitude faster: the file is taken into memory once and plain decimal tokens
// ([+-]digits[.digits][e[+-]digits]) are converted directly.  A decimal with at most 15 significant digits and a
// power of ten up to 10^22 is an integer below 2^53 times or divided by an exactly representable power of ten:
// one IEEE operation, correctly rounded, the value strtod (and with it the stream extraction) returns.  Longer
// tokens go through strtod.  A file that holds anything else than plain decimal tokens is reported as not plain,
// and the caller reads it with a stream, whose treatment of malformed entries is the reference's own.
#ifndef ATOMDEV_TEXT_NUMBERS_H_
#define ATOMDEV_TEXT_NUMBERS_H_

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "host_parallel.h"

class TextNumbers {
public:
    // false: the file could not be opened
    bool open(const char* path) {
        FILE* f = std::fopen(path, "rb");
        if (!f) return false;
        buf_.clear();
        char block[1 << 16];
        for (size_t got; (got = std::fread(block, 1, sizeof block, f)) > 0;) buf_.append(block, got);
        std::fclose(f);
        pos_ = 0;
        // pieces that begin at white space (no token is cut), checked side by side by a few host threads
        cuts_.assign(1, 0);
        const size_t pieces = buf_.size() < (1u << 16) ? 1 : (size_t)writer_threads();
        for (size_t c = 1; c < pieces; c++) {
            size_t p = c * buf_.size() / pieces;
            while (p < buf_.size() && !is_space(buf_[p])) p++;
            if (p > cuts_.back()) cuts_.push_back(p);
        }
        cuts_.push_back(buf_.size());
        plain_ = buf_.find('\0') == std::string::npos;
        if (plain_) {
            std::vector<char> ok(cuts_.size() - 1, 0);
            host_parallel_run(ok.size(), [&](size_t c) { ok[c] = check_plain(cuts_[c], cuts_[c + 1]) ? 1 : 0; });
            for (char o : ok) plain_ = plain_ && o;
        }
        return true;
    }
    bool plain() const { return plain_; }
    // a token read so far was beyond the range of a double: the values are not what a stream would have delivered
    bool out_of_range() const { return out_of_range_; }
    // next number of a plain file; false when the file has no further token
    bool next(double& x) {
        bool oor = false;
        const bool got = parse_at(buf_.c_str(), buf_.size(), pos_, x, oor);
        if (oor) out_of_range_ = true;
        return got;
    }
    // every number of a plain file, in order (a few host threads, each on a piece of the text)
    void all(std::vector<double>& out) {
        const size_t pieces = cuts_.size() - 1;
        std::vector<size_t> count(pieces + 1, 0);
        const char* s = buf_.c_str();
        host_parallel_run(pieces, [&](size_t c) {
            size_t n = 0;
            bool in_token = false;
            for (size_t p = cuts_[c]; p < cuts_[c + 1]; p++) {
                const bool sp = is_space(s[p]);
                if (!sp && !in_token) n++;
                in_token = !sp;
            }
            count[c + 1] = n;
        });
        for (size_t c = 0; c < pieces; c++) count[c + 1] += count[c];
        out.resize(count[pieces]);
        std::vector<char> oor(pieces, 0);
        host_parallel_run(pieces, [&](size_t c) {
            size_t p = cuts_[c], e = count[c];
            bool o = false;
            double x;
            while (e < count[c + 1] && parse_at(s, cuts_[c + 1], p, x, o)) out[e++] = x;
            oor[c] = o ? 1 : 0;
        });
        for (char o : oor) if (o) out_of_range_ = true;
        pos_ = buf_.size();
    }

private:
    // the token at or after `pos` (before `end`), the way the stream extraction converts it; false: no further token
    static bool parse_at(const char* s, size_t end, size_t& pos, double& x, bool& out_of_range) {
        size_t p = pos;
        while (p < end && is_space(s[p])) p++;
        if (p >= end || !s[p]) { pos = p; return false; }
        const size_t start = p;
        bool neg = false;
        if (s[p] == '+' || s[p] == '-') { neg = s[p] == '-'; p++; }
        unsigned long long mant = 0;
        int digits = 0, scale = 0;  // significant digits taken, decimal places among them
        bool lead = true;
        for (; s[p] >= '0' && s[p] <= '9'; p++) {
            if (lead && s[p] == '0') continue;
            lead = false;
            if (digits < 19) mant = mant * 10 + (unsigned)(s[p] - '0');
            digits++;
        }
        if (s[p] == '.') {
            for (p++; s[p] >= '0' && s[p] <= '9'; p++) {
                scale++;
                if (lead && s[p] == '0') continue;
                lead = false;
                if (digits < 19) mant = mant * 10 + (unsigned)(s[p] - '0');
                digits++;
            }
        }
        int e10 = 0;
        if (s[p] == 'e' || s[p] == 'E') {
            p++;
            bool eneg = false;
            if (s[p] == '+' || s[p] == '-') { eneg = s[p] == '-'; p++; }
            for (; s[p] >= '0' && s[p] <= '9'; p++) e10 = e10 < 10000 ? e10 * 10 + (s[p] - '0') : e10;
            if (eneg) e10 = -e10;
        }
        pos = p;
        const int power = e10 - scale;
        if (digits <= 15 && power >= -22 && power <= 22) {
            static const double p10[23] = { 1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
                                            1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22 };
            const double m = (double)mant;
            const double v = power < 0 ? m / p10[-power] : m * p10[power];
            x = neg ? -v : v;
        } else {
            errno = 0;
            x = std::strtod(s + start, nullptr);
            if (errno == ERANGE) out_of_range = true;  // the stream sets its fail bit there and stores a limit value
        }
        return true;
    }
    static bool is_space(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\f' || c == '\v'; }
    // every token: [+-] digits [. digits] [e [+-] digits] with at least one digit in the mantissa and, after an e, in the exponent
    bool check_plain(size_t from, size_t to) const {
        const char* s = buf_.c_str();
        size_t p = from;
        for (;;) {
            while (p < to && is_space(s[p])) p++;
            if (p >= to) return true;
            if (s[p] == '+' || s[p] == '-') p++;
            int nd = 0;
            for (; s[p] >= '0' && s[p] <= '9'; p++) nd++;
            if (s[p] == '.') for (p++; s[p] >= '0' && s[p] <= '9'; p++) nd++;
            if (!nd) return false;
            if (s[p] == 'e' || s[p] == 'E') {
                p++;
                if (s[p] == '+' || s[p] == '-') p++;
                int ne = 0;
                for (; s[p] >= '0' && s[p] <= '9'; p++) ne++;
                if (!ne) return false;
            }
            if (s[p] && !is_space(s[p])) return false;
        }
    }
    std::string buf_;
    std::vector<size_t> cuts_;   // piece boundaries (each at white space)
    size_t pos_ = 0;
    bool plain_ = false, out_of_range_ = false;
};

// The third number of each of the first n triples of a file (lon lat value per entry).  False when the file is not
// plain or holds fewer than n whole triples: the caller then reads it with a stream.  *opened tells whether the file
// exists at all.
inline bool text_third_of_triples(const char* path, size_t n, std::vector<double>& third, bool* opened) {
    TextNumbers t;
    *opened = t.open(path);
    if (!*opened || !t.plain()) return false;
    std::vector<double> every;
    t.all(every);
    if (every.size() < 3 * n) return false;
    third.resize(n);
    for (size_t e = 0; e < n; e++) third[e] = every[3 * e + 2];
    return !t.out_of_range();
}

#endif  // ATOMDEV_TEXT_NUMBERS_H_
