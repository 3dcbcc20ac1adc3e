// BC_Bathymetry_Atmosphere::BC_MountainSurface (atmosphere/BC_Bath_Atm.cpp:39-90): the topography reader of the
// atmosphere.  Host work (a text file); the result is uploaded as ATOMDEV_H and ATOMDEV_TOPOGRAPHY.
#include <algorithm>
#include <cmath>
#include <fstream>
#include <string>
#include <vector>

#include "atomdev_internal.h"
#include "text_numbers.h"

extern "C" int atomdev_atm_read_mountain_surface(atomdev_t* h, const char* path) {
    ATOMDEV_REQUIRE(h && path && h->model == ATOMDEV_ATM, "atomdev_atm_read_mountain_surface: needs an atmosphere handle and a path");
    const int im = h->im, jm = h->jm, km = h->km;
    const double L_atm = h->p.L_atm;
    std::vector<uint8_t> solid(h->n3, 0);  // h as bytes: 1 = solid
    std::vector<double> topography(h->n2, 0.), values;
    auto entry = [&](int j, int k, double height) {
        if (height < 0.) {
            solid[idx3(jm, km, 0, j, k)] = 0;
            topography[(size_t)j * km + k] = 0.;
        } else {
            const int i_h = std::min(int(std::floor(height / L_atm * (im - 1))), im - 1);  // the reference has no upper bound
            topography[(size_t)j * km + k] = height;
            for (int i = 0; i <= i_h; i++) solid[idx3(jm, km, i, j, k)] = 1;
        }
    };
    bool opened = false;
    if (text_third_of_triples(path, h->n2, values, &opened)) {
        for (int j = 0; j < jm; j++)
            for (int k = 0; k < km; k++) entry(j, k, values[(size_t)j * km + k]);
    } else {  // short or malformed file: the stream's own treatment of it, as in the reference
        ATOMDEV_REQUIRE(opened, std::string("could not open Name_Bathymetry_File file: ") + path);
        std::ifstream file(path);
        int j = 0, k = 0;
        for (j = 0; j < jm && !file.eof(); j++)
            for (k = 0; k < km && !file.eof(); k++) {
                double lon, lat, height = -999;  // an entry that is no number counts as sea
                file >> lon >> lat >> height;
                entry(j, k, height);
                if (file.fail()) {
                    file.clear();
                    std::string rest;
                    std::getline(file, rest);
                }
            }
        ATOMDEV_REQUIRE(j == jm && k == km, "wrong topography file size");
    }
    // -180..180 -> 0..360 (lib/Utils.cpp:31-37)
    auto rotate = [km](auto* row) {
        for (int n = 0; n < km / 2; n++) std::swap(row[n], row[km / 2 + n]);
        row[km - 1] = row[0];
    };
    for (int jj = 0; jj < jm; jj++) {
        rotate(&topography[(size_t)jj * km]);
        for (int i = 0; i < im; i++) rotate(&solid[idx3(jm, km, i, jj, 0)]);
    }
    if (int rc = atomdev_upload_mask_bytes(h, solid.data())) return rc;
    return atomdev_upload2d(h, ATOMDEV_TOPOGRAPHY, topography.data());
}
