/*
 * atom_models.hpp - the models' outer interface on top of libatomdev: cAtmosphereModel and cHydrosphereModel with the
 * configuration parameters as public members, LoadConfig over the XML schema, Run and RunTimeSlice.
 *
 * Mirrors atmosphere/cAtmosphereModel.h:27-87 and hydrosphere/cHydrosphereModel.h:15-26 of the reference: the same
 * class names, method names and member names (the members are those param.py:9-178 generates: <common> plus the
 * model's own section, include/atom_params.def), the same defaults, the same treatment of a config file (a missing
 * parameter keeps its default, a missing <atom>, <common> or model section is fatal: cAtmosphereModel.cpp:106-136).
 * What the classes own is different: no Array members, one device handle whose fields stay in HBM; RunTimeSlice is one
 * call of atomdev_atm_run_time_slice / atomdev_hyd_run_time_slice plus the output files.
 *
 * Differences a caller can observe:
 *  - the plate-reconstruction script of a paleo time slice (cAtmosphereModel.cpp:177-189) is started the same way, but a
 *    missing result is an error here instead of an abort inside the surface-file reader;
 *  - the temperature curve is read when it is first needed, from temperature_curve_file as the configuration left it (the
 *    reference reads it in the constructor, before LoadConfig can change the path);
 *  - errors are std::runtime_error exceptions with the library's message, where the reference prints and aborts;
 *  - the 3-D sweeps of both models run in their exact form here (every quotient with the bits of the reference's division,
 *    so that the output files carry the reference's bytes); the faster tolerance forms are chosen with the environment
 *    variable ATOMDEV_RK_VARIANT (2 or 3) or atomdev_set_rk_variant on device_handle(): every field within 1e-10,
 *    single numbers of the text files may then differ in their last printed digit.
 */
#ifndef ATOM_MODELS_HPP_
#define ATOM_MODELS_HPP_

#include <map>
#include <set>
#include <string>

#include "atomdev.h"

namespace atom_detail {  // the parameter table's type names
typedef int t_int;
typedef double t_double;
typedef bool t_bool;
typedef std::string t_string;
}

class cAtmosphereModel {
public:
    cAtmosphereModel();
    ~cAtmosphereModel();
    cAtmosphereModel(const cAtmosphereModel&) = delete;
    cAtmosphereModel& operator=(const cAtmosphereModel&) = delete;

    static cAtmosphereModel* get_model() {
        if (!m_model) m_model = new cAtmosphereModel();
        return m_model;
    }

    void LoadConfig(const char* filename);
    void Run();
    void RunTimeSlice(int time_slice);

    // the time slices run so far (cAtmosphereModel.h:47-74)
    std::set<float>::const_iterator get_current_time() const;
    std::set<float>::const_iterator get_previous_time() const;
    bool is_first_time_slice() const;

    float get_mean_temperature_from_curve(float time) const;

    // The tropopause level of every latitude row, jm entries (cAtmosphereModel.h:80-82): what BC_Thermo::TropopauseLocation
    // (BC_Thermo.cpp:884-899) leaves there during a time slice, tropopause_equator on every row; zeros before the first.
    int* get_tropopause() const { return im_tropopause; }

    // The cosine-of-latitude weighted mean of the temperature at sea level, in Celsius, as a float
    // (cAtmosphereModel.cpp:952-988).  Without an argument: of the model's own temperature field, which is on the device
    // (the reference's private overload, cAtmosphereModel.h:108-110).  With one: of any object that is indexed like the
    // reference's Array, temp.x[0][j][k] (cAtmosphereModel.h:84) - the Array of include/atomdev_mirror.hpp or the caller's own.
    float calculate_mean_temperature();
    template <class ArrayLike>
    float calculate_mean_temperature(const ArrayLike& temp) const {
        double ret = 0., weight = 0.;
        for (int j = 0; j < jm; j++) {
            const double w = node_weight(j);
            for (int k = 0; k < km; k++) {
                ret += temp.x[0][j][k] * w;
                weight += w;
            }
        }
        return (float)((ret / weight - 1) * t_0);
    }

    // the configuration parameters, as the reference's AtmosphereParams.h.inc declares them
#define ATOM_PARAM_common(type, name) atom_detail::t_##type name;
#define ATOM_PARAM_atmosphere(type, name) atom_detail::t_##type name;
#define ATOM_PARAM_hydrosphere(type, name)
#define ATOM_PARAM(section, type, name, dflt) ATOM_PARAM_##section(type, name)
#include "atom_params.def"
#undef ATOM_PARAM
#undef ATOM_PARAM_common
#undef ATOM_PARAM_atmosphere
#undef ATOM_PARAM_hydrosphere

    static const double pi180, the_degree, phi_degree, dthe, dphi, dr, dt;
    static const double the0, phi0, r0;
    static const int im = 41, jm = 181, km = 361, nm = 200;

    // the device handle that holds the model's fields (created by the first RunTimeSlice; NULL before)
    atomdev_t* device_handle() const { return m_handle; }
    // what the last RunTimeSlice returned: the carried BC_Thermo members and the last residuum pair
    const atomdev_atm_loop_state& last_state() const { return m_state; }

private:
    void SetDefaultConfig();
    void load_temperature_curve() const;
    static double node_weight(int j);   // calculate_node_weights (:971-988): the cosine of the row's latitude

    int* im_tropopause;

    static cAtmosphereModel* m_model;
    mutable std::map<float, float> m_temperature_curve;
    mutable bool m_curve_loaded;
    std::set<float> m_time_list;
    std::set<float>::const_iterator m_current_time;
    atomdev_t* m_handle;
    atomdev_atm_loop_state m_state;
};

class cHydrosphereModel {
public:
    cHydrosphereModel();
    ~cHydrosphereModel();
    cHydrosphereModel(const cHydrosphereModel&) = delete;
    cHydrosphereModel& operator=(const cHydrosphereModel&) = delete;

    void LoadConfig(const char* filename);
    void Run();
    void RunTimeSlice(int time_slice);

#define ATOM_PARAM_common(type, name) atom_detail::t_##type name;
#define ATOM_PARAM_atmosphere(type, name)
#define ATOM_PARAM_hydrosphere(type, name) atom_detail::t_##type name;
#define ATOM_PARAM(section, type, name, dflt) ATOM_PARAM_##section(type, name)
#include "atom_params.def"
#undef ATOM_PARAM
#undef ATOM_PARAM_common
#undef ATOM_PARAM_atmosphere
#undef ATOM_PARAM_hydrosphere

    static const int im = 41, jm = 181, km = 361, nm = 200;

    atomdev_t* device_handle() const { return m_handle; }
    // continuity residuum before and after the last velocity iteration of the last RunTimeSlice
    const double* last_residuum() const { return m_residuum; }

private:
    void SetDefaultConfig();
    atomdev_t* m_handle;
    double m_residuum[2];
};

#endif  // ATOM_MODELS_HPP_
