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
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <string>

class TextNumbers {
public:
    // false: the file could not be opened
    bool open(const char* path) {
        std::ifstream f(path, std::ios::binary);
        if (!f.is_open()) return false;
        buf_.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
        pos_ = 0;
        plain_ = check_plain();
        return true;
    }
    bool plain() const { return plain_; }
    // a token read so far was beyond the range of a double: the values are not what a stream would have delivered
    bool out_of_range() const { return out_of_range_; }
    // next number of a plain file; false when the file has no further token
    bool next(double& x) {
        const char* s = buf_.c_str();
        size_t p = pos_;
        while (is_space(s[p])) p++;
        if (!s[p]) { pos_ = p; return false; }
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
        pos_ = p;
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
            if (errno == ERANGE) out_of_range_ = true;  // the stream sets its fail bit there and stores a limit value
        }
        return true;
    }

private:
    static bool is_space(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\f' || c == '\v'; }
    // every token: [+-] digits [. digits] [e [+-] digits] with at least one digit in the mantissa and, after an e, in the exponent
    bool check_plain() const {
        const char* s = buf_.c_str();
        if (buf_.find('\0') != std::string::npos) return false;
        size_t p = 0;
        for (;;) {
            while (is_space(s[p])) p++;
            if (!s[p]) return true;
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
    size_t pos_ = 0;
    bool plain_ = false, out_of_range_ = false;
};

// The third number of each of the first n triples of a file (lon lat value per entry).  False when the file is not
// plain or holds fewer than n whole triples: the caller then reads it with a stream.  *opened tells whether the file
// exists at all.
#include <vector>
inline bool text_third_of_triples(const char* path, size_t n, std::vector<double>& third, bool* opened) {
    TextNumbers t;
    *opened = t.open(path);
    if (!*opened || !t.plain()) return false;
    third.resize(n);
    double lon, lat;
    for (size_t e = 0; e < n; e++)
        if (!t.next(lon) || !t.next(lat) || !t.next(third[e])) return false;
    return !t.out_of_range();
}

#endif  // ATOMDEV_TEXT_NUMBERS_H_
