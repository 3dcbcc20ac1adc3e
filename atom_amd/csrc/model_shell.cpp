// The models' outer interface (include/atom_models.hpp): configuration members, LoadConfig over a small XML reader,
// Run / RunTimeSlice on top of the C-ABI of this library, and the flat C interface the Python module binds.
// Host code only; everything the device does goes through the entry points of include/atomdev.h.
#include <sys/stat.h>
#include <sys/types.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "atom_models.hpp"
#include "atom_models_c.h"

using std::string;

namespace {

// ---- a reader for the part of XML the configuration files use ------------------------------------------------
// Elements, text, comments, a prolog, attributes (skipped), the five predefined entities.  An element's text is the
// text of its first child node when that node is text, as tinyxml2's GetText() reports it; a child is looked up by
// name, the first match wins (FirstChildElement).
struct XmlNode {
    string name;
    bool first_is_text = false;
    string text;
    std::vector<std::unique_ptr<XmlNode>> children;
    const XmlNode* child(const char* n) const {
        for (const auto& c : children)
            if (c->name == n) return c.get();
        return nullptr;
    }
};

class XmlReader {
public:
    explicit XmlReader(const string& s) : s_(s), p_(0) {}
    std::unique_ptr<XmlNode> document() {
        auto root = std::make_unique<XmlNode>();
        content(*root, "");
        return root;
    }

private:
    const string& s_;
    size_t p_;
    [[noreturn]] void fail(const string& what) const {
        size_t line = 1;
        for (size_t i = 0; i < p_ && i < s_.size(); i++) line += s_[i] == '\n';
        throw std::invalid_argument("XML: " + what + " in line " + std::to_string(line));
    }
    bool starts(const char* lit) const { return s_.compare(p_, strlen(lit), lit) == 0; }
    void skip_to(const char* lit) {
        const size_t q = s_.find(lit, p_);
        if (q == string::npos) fail(string("missing '") + lit + "'");
        p_ = q + strlen(lit);
    }
    static string unescape(const string& t) {
        static const struct { const char* ent; char c; } E[] = { { "&lt;", '<' }, { "&gt;", '>' }, { "&amp;", '&' }, { "&quot;", '"' }, { "&apos;", '\'' } };
        string out;
        for (size_t i = 0; i < t.size();) {
            bool hit = false;
            if (t[i] == '&')
                for (const auto& e : E)
                    if (t.compare(i, strlen(e.ent), e.ent) == 0) { out += e.c; i += strlen(e.ent); hit = true; break; }
            if (!hit) out += t[i++];
        }
        return out;
    }
    string name() {
        const size_t b = p_;
        while (p_ < s_.size() && (isalnum((unsigned char)s_[p_]) || s_[p_] == '_' || s_[p_] == '-' || s_[p_] == ':' || s_[p_] == '.')) p_++;
        if (p_ == b) fail("element name expected");
        return s_.substr(b, p_ - b);
    }
    // children and text of `node` up to its closing tag (or the end of input for the document)
    void content(XmlNode& node, const string& closing) {
        bool first = true;
        for (;;) {
            const size_t lt = s_.find('<', p_);
            const string raw = s_.substr(p_, lt == string::npos ? string::npos : lt - p_);
            const bool blank = raw.find_first_not_of(" \t\r\n") == string::npos;
            if (!blank) {  // whitespace between elements is no node
                if (first) { node.first_is_text = true; node.text = unescape(raw); }
                first = false;
            }
            if (lt == string::npos) {
                if (!closing.empty()) fail("missing </" + closing + ">");
                p_ = s_.size();
                return;
            }
            p_ = lt;
            if (starts("<!--")) { skip_to("-->"); first = false; continue; }  // a comment is a node of its own
            if (starts("<?")) { skip_to("?>"); continue; }
            if (starts("<!")) { skip_to(">"); continue; }
            if (starts("</")) {
                p_ += 2;
                const string n = name();
                if (n != closing) fail("</" + n + "> closes <" + closing + ">");
                skip_to(">");
                return;
            }
            p_++;
            auto child = std::make_unique<XmlNode>();
            child->name = name();
            // attributes are not used by the configuration: skipped, quotes respected
            bool self_closing = false;
            for (;;) {
                if (p_ >= s_.size()) fail("unterminated tag <" + child->name);
                const char c = s_[p_];
                if (c == '"' || c == '\'') {
                    const size_t q = s_.find(c, p_ + 1);
                    if (q == string::npos) fail("unterminated attribute value");
                    p_ = q + 1;
                } else if (c == '/' && p_ + 1 < s_.size() && s_[p_ + 1] == '>') { self_closing = true; p_ += 2; break; }
                else if (c == '>') { p_++; break; }
                else p_++;
            }
            if (!self_closing) content(*child, child->name);
            node.children.push_back(std::move(child));
            first = false;
        }
    }
};

string read_file(const char* filename) {
    std::ifstream f(filename, std::ios::binary);
    if (!f.is_open()) throw std::invalid_argument(string("unable to load config file:  ") + filename);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

// Config::Fill*WithElement (lib/Config.cpp:6-68): an absent element or an element without text keeps the destination
void fill(const XmlNode* parent, const char* name, double& dest) {
    const XmlNode* e = parent->child(name);
    if (!e || !e->first_is_text) return;
    try { dest = std::stod(e->text); }
    catch (const std::exception&) { throw std::invalid_argument("while reading XML file, could not convert '" + e->text + "' to double for parameter " + name); }
}
void fill(const XmlNode* parent, const char* name, int& dest) {
    const XmlNode* e = parent->child(name);
    if (!e || !e->first_is_text) return;
    try { dest = std::stoi(e->text); }
    catch (const std::exception&) { throw std::invalid_argument("while reading XML file, could not convert '" + e->text + "' to int for parameter " + name); }
}
void fill(const XmlNode* parent, const char* name, string& dest) {
    const XmlNode* e = parent->child(name);
    if (!e || !e->first_is_text) return;
    dest = e->text;
}
void fill(const XmlNode* parent, const char* name, bool& dest) {
    const XmlNode* e = parent->child(name);
    if (!e) return;
    if (e->first_is_text && e->text == "false") dest = false;
    else if (e->first_is_text && e->text == "true") dest = true;
    else throw std::invalid_argument("while reading XML file, could not convert '" + e->text + "' to bool (true/false) for parameter " + name);
}

// The sections a model reads, by the reference's messages (cAtmosphereModel.cpp:116-128)
void open_config(const char* filename, const char* section, std::unique_ptr<XmlNode>& doc, const XmlNode*& common, const XmlNode*& own) {
    const string text = read_file(filename);
    doc = XmlReader(text).document();
    const XmlNode* atom = doc->child("atom");
    if (!atom) throw std::invalid_argument(string("Failed to find the 'atom' element in config file: ") + filename);
    common = atom->child("common");
    if (!common) throw std::invalid_argument(string("Failed to find the 'common' element in 'atom' element in config file: ") + filename);
    own = atom->child(section);
    if (!own) throw std::invalid_argument(string("Failed to find the '") + section + "' element in 'atom' element in config file: " + filename);
}

void check(int rc, const char* what) {
    if (rc != 0) throw std::runtime_error(string(what) + ": " + atomdev_last_error());
}

bool file_exists(const string& path) {
    struct stat info;
    return stat(path.c_str(), &info) == 0;
}

void make_directory(const string& path) {
    if (!file_exists(path)) mkdir(path.c_str(), 0777);
}

}  // namespace

// ==== cAtmosphereModel ========================================================================================

cAtmosphereModel* cAtmosphereModel::m_model = nullptr;
const double cAtmosphereModel::pi180 = 180. / M_PI;
const double cAtmosphereModel::the_degree = 1.;
const double cAtmosphereModel::phi_degree = 1.;
const double cAtmosphereModel::dthe = the_degree / pi180;  // cAtmosphereModel.cpp:38-55
const double cAtmosphereModel::dphi = phi_degree / pi180;
const double cAtmosphereModel::dr = 0.025;
const double cAtmosphereModel::dt = 0.00001;
const double cAtmosphereModel::the0 = 0.;
const double cAtmosphereModel::phi0 = 0.;
const double cAtmosphereModel::r0 = 1.;

cAtmosphereModel::cAtmosphereModel() : m_curve_loaded(false), m_handle(nullptr) {
    im_tropopause = new int[jm]();  // cAtmosphereModel.cpp:84
    SetDefaultConfig();
    memset(&m_state, 0, sizeof m_state);
    m_model = this;
}

cAtmosphereModel::~cAtmosphereModel() {
    delete[] im_tropopause;
    if (m_handle) atomdev_destroy(m_handle);
    if (m_model == this) m_model = nullptr;
}

void cAtmosphereModel::SetDefaultConfig() {
#define ATOM_PARAM_common(name, dflt) name = dflt;
#define ATOM_PARAM_atmosphere(name, dflt) name = dflt;
#define ATOM_PARAM_hydrosphere(name, dflt)
#define ATOM_PARAM(section, type, name, dflt) ATOM_PARAM_##section(name, dflt)
#include "atom_params.def"
#undef ATOM_PARAM
#undef ATOM_PARAM_common
#undef ATOM_PARAM_atmosphere
#undef ATOM_PARAM_hydrosphere
}

void cAtmosphereModel::LoadConfig(const char* filename) {
    std::unique_ptr<XmlNode> doc;
    const XmlNode *common = nullptr, *own = nullptr;
    open_config(filename, "atmosphere", doc, common, own);
#define ATOM_PARAM_common(name) fill(common, #name, name);
#define ATOM_PARAM_atmosphere(name) fill(own, #name, name);
#define ATOM_PARAM_hydrosphere(name)
#define ATOM_PARAM(section, type, name, dflt) ATOM_PARAM_##section(name)
#include "atom_params.def"
#undef ATOM_PARAM
#undef ATOM_PARAM_common
#undef ATOM_PARAM_atmosphere
#undef ATOM_PARAM_hydrosphere
    m_curve_loaded = false;  // temperature_curve_file may have changed
}

std::set<float>::const_iterator cAtmosphereModel::get_current_time() const {
    if (m_time_list.empty()) throw std::runtime_error("The time list is empty. It is likely the model has not started yet.");
    return m_current_time;
}
std::set<float>::const_iterator cAtmosphereModel::get_previous_time() const {
    if (m_time_list.empty()) throw std::runtime_error("The time list is empty. It is likely the model has not started yet.");
    if (m_current_time == m_time_list.begin())
        throw std::runtime_error("The current time is the only time slice for now. There is no previous time yet.");
    std::set<float>::const_iterator ret = m_current_time;
    return --ret;
}
bool cAtmosphereModel::is_first_time_slice() const {
    if (m_time_list.empty()) throw std::runtime_error("The time list is empty. It is likely the model has not started yet.");
    return m_current_time == m_time_list.begin();
}

// cAtmosphereModel::load_temperature_curve (cAtmosphereModel.cpp:904-919): "time temperature" per line, as floats
void cAtmosphereModel::load_temperature_curve() const {
    m_temperature_curve.clear();
    std::ifstream f(temperature_curve_file);
    if (!f.is_open()) throw std::runtime_error("error while opening file: " + temperature_curve_file);
    string line;
    float time = 0.f, temperature = 0.f;
    while (std::getline(f, line)) {
        std::stringstream(line) >> time >> temperature;
        m_temperature_curve.insert(std::pair<float, float>(time, temperature));
    }
    m_curve_loaded = true;
}

// cAtmosphereModel::get_mean_temperature_from_curve (:924-946): linear between the two entries around `time`, in float
float cAtmosphereModel::get_mean_temperature_from_curve(float time) const {
    if (!m_curve_loaded) load_temperature_curve();
    const std::map<float, float>& curve = m_temperature_curve;
    if (curve.size() < 2) return NAN;
    if (time < curve.begin()->first || time > (--curve.end())->first) return NAN;
    std::map<float, float>::const_iterator upper = curve.begin(), bottom = ++curve.begin();
    for (std::map<float, float>::const_iterator it = curve.begin(); it != curve.end(); ++it) {
        if (time < it->first) { bottom = it; break; }
        upper = it;
    }
    return upper->second + (time - upper->first) / (bottom->first - upper->first) * (bottom->second - upper->second);
}

void cAtmosphereModel::Run() {
    make_directory(output_path);
    for (int i = time_start; i <= time_end; i += time_step) RunTimeSlice(i);
}

void cAtmosphereModel::RunTimeSlice(int Ma) {
    m_current_time = m_time_list.insert(float(Ma)).first;  // :149
    make_directory(output_path);
    // the surface files; a paleo time slice takes them from the plate reconstruction (:174-189)
    string temperature = temperature_file, precipitation = precipitation_file;
    if (Ma != 0 && use_earthbyte_reconstruction) {
        temperature = output_path + "/" + std::to_string(Ma) + "Ma_Reconstructed_Temperature.xyz";
        precipitation = output_path + "/" + std::to_string(Ma) + "Ma_Reconstructed_Precipitation.xyz";
        if (!file_exists(temperature) || !file_exists(precipitation)) {
            const string cmd = "python " + reconstruction_script_path + " " + std::to_string(Ma - time_step) + " " + std::to_string(Ma) + " " +
                               output_path + " " + BathymetrySuffix + " atm";
            const int ret = system(cmd.c_str());
            if (verbose) std::cout << " reconstruction script returned: " << ret << std::endl;
            if (!file_exists(temperature) || !file_exists(precipitation))
                throw std::runtime_error("the reconstruction script left no " + temperature + " / " + precipitation + " (" + cmd + ")");
        }
    }
    const string bathymetry_name = std::to_string(Ma) + BathymetrySuffix;
    const string bathymetry_filepath = bathymetry_path + "/" + bathymetry_name;

    atomdev_params P;
    atomdev_default_params(ATOMDEV_ATM, &P);
    P.dt = dt; P.dr = dr; P.dthe = dthe; P.dphi = dphi; P.r0 = r0; P.the0 = the0; P.phi0 = phi0;
    P.re = re; P.pr = pr; P.g = g; P.Buoyancy = Buoyancy; P.u_0 = u_0; P.t_0 = t_0; P.c_0 = c_0;
    P.sc_WaterVapour = sc_WaterVapour; P.sc_CO2 = sc_CO2; P.gam = gam; P.r_air = r_air; P.R_Air = R_Air; P.R_WaterVapour = R_WaterVapour;
    P.L_atm = L_atm; P.cp_l = cp_l; P.t_tropopause = t_tropopause; P.ta = ta; P.pa = pa; P.ca = ca; P.p_0 = p_0;
    P.ep = ep; P.hp = hp; P.lv = lv; P.ls = ls; P.r_water_vapour = r_water_vapour; P.co2_0 = co2_0;
    if (!m_handle) {
        check(atomdev_create(ATOMDEV_ATM, im, jm, km, &P, -1, &m_handle), "atomdev_create");
        // the models' files are the reference's bytes unless the caller asks for a faster form of the sweep
        if (!getenv("ATOMDEV_RK_VARIANT")) check(atomdev_set_rk_variant(m_handle, 0), "atomdev_set_rk_variant");
    } else check(atomdev_set_params(m_handle, &P), "atomdev_set_params");

    atomdev_atm_slice S;
    atomdev_default_atm_slice(&S);
    atomdev_atm_start& st = S.start;
    st.Ma = int(std::round(*get_current_time()));
    st.first_time_slice = is_first_time_slice() ? 1 : 0;
    st.Ma_previous = st.first_time_slice ? 0 : int(std::round(*get_previous_time()));
    st.sun = sun; st.RadiationModel = RadiationModel; st.NASATemperature = NASATemperature; st.declination = declination;
    st.sun_position_lat = sun_position_lat; st.sun_position_lon = sun_position_lon;
    // BC_Thermo.cpp:93-94, 446-451: differences of floats
    st.t_cretaceous = get_mean_temperature_from_curve(st.Ma) - get_mean_temperature_from_curve(0);
    st.t_cretaceous_add = 0.;
    if (!st.first_time_slice) {
        st.t_cretaceous_add = get_mean_temperature_from_curve(st.Ma) - get_mean_temperature_from_curve(*get_previous_time());
        st.t_cretaceous_add /= t_0;
    }
    st.t_average = t_average; st.t_equator = t_equator; st.t_pole = t_pole; st.t_land = t_land;
    st.c_tropopause = c_tropopause; st.c_ocean = c_ocean; st.c_land = c_land;
    st.co2_average = co2_average; st.co2_equator = co2_equator; st.co2_tropopause = co2_tropopause; st.co2_pole = co2_pole;
    st.co2_cretaceous = co2_cretaceous; st.co2_vegetation = co2_vegetation; st.co2_ocean = co2_ocean; st.co2_land = co2_land;
    for (int j = 0; j < jm; j++) im_tropopause[j] = tropopause_equator;  // BC_Thermo::TropopauseLocation (:884-899)
    st.im_tropopause = im_tropopause;
    atomdev_atm_loop& L = S.loop;
    L.velocity_iter_max_2D = velocity_iter_max_2D; L.pressure_iter_max_2D = pressure_iter_max_2D;
    L.velocity_iter_max = velocity_iter_max; L.pressure_iter_max = pressure_iter_max;
    L.RadiationModel = RadiationModel; L.queries = 1; L.nm = nm;
    L.radiation.albedo_pole = albedo_pole; L.radiation.albedo_equator = albedo_equator; L.radiation.epsilon_pole = epsilon_pole;
    L.radiation.epsilon_equator = epsilon_equator; L.radiation.epsilon_tropopause = epsilon_tropopause;
    L.radiation.rad_pole = rad_pole; L.radiation.rad_equator = rad_equator; L.radiation.sigma = sigma;
    S.tropopause_equator = tropopause_equator;
    S.coast_correction = (NASATemperature == 1 && Ma > 0 && !use_earthbyte_reconstruction) ? 1 : 0;  // :276
    S.checkpoint = checkpoint;
    S.CO2 = CO2; S.co2_factor = co2_factor;
    S.ua = ua; S.va = va; S.wa = wa; S.coa = coa;
    S.mean_temperature = get_mean_temperature_from_curve(*get_current_time());  // :994
    S.bathymetry_file = bathymetry_filepath.c_str();
    S.temperature_file = temperature.c_str();
    S.precipitation_file = precipitation.c_str();
    S.output_path = output_path.c_str();
    S.bathymetry_name = bathymetry_name.c_str();
    if (verbose) std::cout << "\n   Output is being written to " << output_path << "\n   Ma = " << Ma << "\n   bathymetry_filepath = " << bathymetry_filepath << "\n\n";
    check(atomdev_atm_run_time_slice(m_handle, &S, &m_state), "atomdev_atm_run_time_slice");
}

double cAtmosphereModel::node_weight(int j) {
    return j <= 90 ? std::cos((90 - j) * M_PI / 180.0) : std::cos((j - 90) * M_PI / 180.0);
}

float cAtmosphereModel::calculate_mean_temperature() {
    if (!m_handle) throw std::runtime_error("calculate_mean_temperature: no time slice has been run yet");
    float mean = 0.f;
    check(atomdev_atm_calculate_mean_temperature(m_handle, &mean), "atomdev_atm_calculate_mean_temperature");
    return mean;
}

// ==== cHydrosphereModel =======================================================================================

cHydrosphereModel::cHydrosphereModel() : m_handle(nullptr) {
    SetDefaultConfig();
    m_residuum[0] = m_residuum[1] = 0.;
}

cHydrosphereModel::~cHydrosphereModel() {
    if (m_handle) atomdev_destroy(m_handle);
}

void cHydrosphereModel::SetDefaultConfig() {
#define ATOM_PARAM_common(name, dflt) name = dflt;
#define ATOM_PARAM_atmosphere(name, dflt)
#define ATOM_PARAM_hydrosphere(name, dflt) name = dflt;
#define ATOM_PARAM(section, type, name, dflt) ATOM_PARAM_##section(name, dflt)
#include "atom_params.def"
#undef ATOM_PARAM
#undef ATOM_PARAM_common
#undef ATOM_PARAM_atmosphere
#undef ATOM_PARAM_hydrosphere
}

void cHydrosphereModel::LoadConfig(const char* filename) {
    std::unique_ptr<XmlNode> doc;
    const XmlNode *common = nullptr, *own = nullptr;
    open_config(filename, "hydrosphere", doc, common, own);
#define ATOM_PARAM_common(name) fill(common, #name, name);
#define ATOM_PARAM_atmosphere(name)
#define ATOM_PARAM_hydrosphere(name) fill(own, #name, name);
#define ATOM_PARAM(section, type, name, dflt) ATOM_PARAM_##section(name)
#include "atom_params.def"
#undef ATOM_PARAM
#undef ATOM_PARAM_common
#undef ATOM_PARAM_atmosphere
#undef ATOM_PARAM_hydrosphere
}

void cHydrosphereModel::Run() {
    make_directory(output_path);
    for (int i = time_start; i <= time_end; i += time_step) RunTimeSlice(i);
}

void cHydrosphereModel::RunTimeSlice(int Ma) {
    make_directory(output_path);
    const double pi180 = 180. / M_PI;
    atomdev_params P;  // cHydrosphereModel.cpp:145-156
    atomdev_default_params(ATOMDEV_HYD, &P);
    P.dt = 0.00001; P.dr = 0.025; P.dthe = 1. / pi180; P.dphi = 1. / pi180; P.r0 = 1.; P.the0 = 0.; P.phi0 = 0.;
    P.re = re; P.pr = pr; P.g = g; P.Buoyancy = Buoyancy; P.u_0 = u_0; P.t_0 = t_0; P.c_0 = c_0;
    P.sc = sc; P.L_hyd = L_hyd; P.r_0_water = r_0_water; P.ta = ta; P.pa = pa; P.ca = ca; P.p_0 = p_0;
    if (!m_handle) {
        check(atomdev_create(ATOMDEV_HYD, im, jm, km, &P, -1, &m_handle), "atomdev_create");
        // the models' files are the reference's bytes unless the caller asks for a faster form of the sweep
        if (!getenv("ATOMDEV_RK_VARIANT")) check(atomdev_set_rk_variant(m_handle, 0), "atomdev_set_rk_variant");
    } else check(atomdev_set_params(m_handle, &P), "atomdev_set_params");
    atomdev_t* h = m_handle;

    // reset_arrays (:636-694): t and c at ta and ca, the velocities at ua, va, wa, the pressures at pa, the rest 0
    for (int f = 0; f < ATOMDEV_N3D; f++) {
        if (f == ATOMDEV_RESIDUUM) continue;
        double v = 0.;
        switch (f) {
        case ATOMDEV_T: case ATOMDEV_TN: v = ta; break;
        case ATOMDEV_U: case ATOMDEV_UN: v = ua; break;
        case ATOMDEV_V: case ATOMDEV_VN: v = va; break;
        case ATOMDEV_W: case ATOMDEV_WN: v = wa; break;
        case ATOMDEV_C: case ATOMDEV_CN: v = ca; break;
        case ATOMDEV_P_DYN: case ATOMDEV_P_DYNN: case ATOMDEV_P_STAT: v = pa; break;
        default: break;
        }
        if (atomdev_fill(h, f, v) != 0) continue;  // a field the hydrosphere does not have
    }
    for (int g2 = 0; g2 < ATOMDEV_N2D; g2++) check(atomdev_fill2d(h, g2, 0.), "atomdev_fill2d");

    const string bathymetry_name = std::to_string(Ma) + BathymetrySuffix;
    const string bathymetry_filepath = bathymetry_path + "/" + bathymetry_name;
    string salinity = salinity_file;
    if (Ma != 0 && use_earthbyte_reconstruction) {  // :192-201
        salinity = output_path + "/" + std::to_string(Ma) + "Ma_Reconstructed_Salinity.xyz";
        const string cmd = "python " + reconstruction_script_path + " " + std::to_string(Ma - time_step) + " " + std::to_string(Ma) + " " +
                           output_path + " " + BathymetrySuffix + " hyd";
        const int ret = system(cmd.c_str());
        if (verbose) std::cout << " reconstruction script returned: " << ret << std::endl;
        if (!file_exists(salinity)) throw std::runtime_error("the reconstruction script left no " + salinity + " (" + cmd + ")");
    }
    const string transfer = output_path + "/[" + bathymetry_name + "]_Transfer_Atm.vw";  // PostProcess_Hyd.cpp:646

    atomdev_hyd_slice S;
    atomdev_default_hyd_slice(im, &S);
    S.start.i_beg = 0; S.start.i_max = im - 1; S.start.Ma = Ma; S.start.Ma_max = 300; S.start.Ma_max_half = 150;  // :140-141, 180-181
    S.start.t_average = t_average; S.start.t_cretaceous_max = t_cretaceous_max;
    S.velocity_iter_max_2D = velocity_iter_max_2D; S.pressure_iter_max_2D = pressure_iter_max_2D;
    S.velocity_iter_max = velocity_iter_max; S.pressure_iter_max = pressure_iter_max;
    S.transfer_file = transfer.c_str();
    S.bathymetry_file = bathymetry_filepath.c_str();
    const bool read_salinity = Ma == 0 || use_earthbyte_reconstruction;  // :273
    S.salinity_file = read_salinity ? salinity.c_str() : nullptr;
    S.queries = 1;
    // write_file (:739-772) at the checkpoints of the loop and at the end, inside the call
    S.checkpoint = checkpoint; S.paraview_panorama_vts = paraview_panorama_vts ? 1 : 0;
    S.output_path = output_path.c_str(); S.bathymetry_name = bathymetry_name.c_str();
    int loc[3];
    check(atomdev_hyd_run_time_slice(h, &S, m_residuum, loc), "atomdev_hyd_run_time_slice");

}

// ==== the flat interface (include/atom_models_c.h) ============================================================

namespace {

thread_local string g_model_error;

struct ParamInfo {
    const char *section, *type, *name;
};
const ParamInfo ATM_PARAMS[] = {
#define ATOM_PARAM_common(type, name) { "common", #type, #name },
#define ATOM_PARAM_atmosphere(type, name) { "atmosphere", #type, #name },
#define ATOM_PARAM_hydrosphere(type, name)
#define ATOM_PARAM(section, type, name, dflt) ATOM_PARAM_##section(type, name)
#include "atom_params.def"
#undef ATOM_PARAM
#undef ATOM_PARAM_common
#undef ATOM_PARAM_atmosphere
#undef ATOM_PARAM_hydrosphere
};
const ParamInfo HYD_PARAMS[] = {
#define ATOM_PARAM_common(type, name) { "common", #type, #name },
#define ATOM_PARAM_atmosphere(type, name)
#define ATOM_PARAM_hydrosphere(type, name) { "hydrosphere", #type, #name },
#define ATOM_PARAM(section, type, name, dflt) ATOM_PARAM_##section(type, name)
#include "atom_params.def"
#undef ATOM_PARAM
#undef ATOM_PARAM_common
#undef ATOM_PARAM_atmosphere
#undef ATOM_PARAM_hydrosphere
};

string to_text(double v) { char b[40]; snprintf(b, sizeof b, "%.17g", v); return b; }
string to_text(int v) { return std::to_string(v); }
string to_text(bool v) { return v ? "true" : "false"; }
string to_text(const string& v) { return v; }
void from_text(const string& t, double& v) { v = std::stod(t); }
void from_text(const string& t, int& v) { v = std::stoi(t); }
void from_text(const string& t, bool& v) {
    if (t == "true") v = true;
    else if (t == "false") v = false;
    else throw std::invalid_argument("'" + t + "' is neither true nor false");
}
void from_text(const string& t, string& v) { v = t; }

template <class M> bool access_atm(M* m, const string& key, string* get, const string* set) {
#define ATOM_PARAM_common(name) if (key == #name) { if (get) *get = to_text(m->name); if (set) from_text(*set, m->name); return true; }
#define ATOM_PARAM_atmosphere(name) ATOM_PARAM_common(name)
#define ATOM_PARAM_hydrosphere(name)
#define ATOM_PARAM(section, type, name, dflt) ATOM_PARAM_##section(name)
#include "atom_params.def"
#undef ATOM_PARAM
#undef ATOM_PARAM_atmosphere
#undef ATOM_PARAM_hydrosphere
    return false;
}
template <class M> bool access_hyd(M* m, const string& key, string* get, const string* set) {
#define ATOM_PARAM_atmosphere(name)
#define ATOM_PARAM_hydrosphere(name) ATOM_PARAM_common(name)
#define ATOM_PARAM(section, type, name, dflt) ATOM_PARAM_##section(name)
#include "atom_params.def"
#undef ATOM_PARAM
#undef ATOM_PARAM_common
#undef ATOM_PARAM_atmosphere
#undef ATOM_PARAM_hydrosphere
    return false;
}

}  // namespace

struct atom_model {
    int kind;
    cAtmosphereModel* atm;
    cHydrosphereModel* hyd;
};

#define GUARD(body)                                             \
    try { body; return 0; }                                     \
    catch (const std::exception& e) { g_model_error = e.what(); return 1; }

extern "C" {

const char* atom_model_last_error(void) { return g_model_error.c_str(); }

atom_model_t* atom_model_create(int kind) {
    if (kind != ATOMDEV_ATM && kind != ATOMDEV_HYD) { g_model_error = "atom_model_create: unknown model"; return nullptr; }
    atom_model* m = new atom_model{ kind, nullptr, nullptr };
    if (kind == ATOMDEV_ATM) m->atm = new cAtmosphereModel();
    else m->hyd = new cHydrosphereModel();
    return m;
}

void atom_model_destroy(atom_model_t* m) {
    if (!m) return;
    delete m->atm;
    delete m->hyd;
    delete m;
}

int atom_model_load_config(atom_model_t* m, const char* filename) {
    if (!m || !filename) { g_model_error = "atom_model_load_config: null argument"; return 1; }
    GUARD(m->atm ? m->atm->LoadConfig(filename) : m->hyd->LoadConfig(filename))
}
int atom_model_run(atom_model_t* m) {
    if (!m) { g_model_error = "atom_model_run: null argument"; return 1; }
    GUARD(m->atm ? m->atm->Run() : m->hyd->Run())
}
int atom_model_run_time_slice(atom_model_t* m, int Ma) {
    if (!m) { g_model_error = "atom_model_run_time_slice: null argument"; return 1; }
    GUARD(m->atm ? m->atm->RunTimeSlice(Ma) : m->hyd->RunTimeSlice(Ma))
}

int atom_model_param_count(int kind) {
    return kind == ATOMDEV_ATM ? (int)(sizeof ATM_PARAMS / sizeof ATM_PARAMS[0]) : kind == ATOMDEV_HYD ? (int)(sizeof HYD_PARAMS / sizeof HYD_PARAMS[0]) : 0;
}
static const ParamInfo* info(int kind, int index) {
    if (index < 0 || index >= atom_model_param_count(kind)) return nullptr;
    return kind == ATOMDEV_ATM ? &ATM_PARAMS[index] : &HYD_PARAMS[index];
}
const char* atom_model_param_name(int kind, int index) { const ParamInfo* p = info(kind, index); return p ? p->name : nullptr; }
const char* atom_model_param_type(int kind, int index) { const ParamInfo* p = info(kind, index); return p ? p->type : nullptr; }
const char* atom_model_param_section(int kind, int index) { const ParamInfo* p = info(kind, index); return p ? p->section : nullptr; }

int atom_model_get_param(atom_model_t* m, const char* name, char* buffer, size_t size) {
    if (!m || !name || !buffer || size == 0) { g_model_error = "atom_model_get_param: null argument"; return 1; }
    string text;
    const bool found = m->atm ? access_atm(m->atm, name, &text, nullptr) : access_hyd(m->hyd, name, &text, nullptr);
    if (!found) { g_model_error = string("no parameter named ") + name; return 1; }
    if (text.size() + 1 > size) { g_model_error = string("atom_model_get_param: buffer too small for ") + name; return 2; }
    memcpy(buffer, text.c_str(), text.size() + 1);
    return 0;
}
int atom_model_set_param(atom_model_t* m, const char* name, const char* text) {
    if (!m || !name || !text) { g_model_error = "atom_model_set_param: null argument"; return 1; }
    const string t = text;
    try {
        const bool found = m->atm ? access_atm(m->atm, name, nullptr, &t) : access_hyd(m->hyd, name, nullptr, &t);
        if (!found) { g_model_error = string("no parameter named ") + name; return 1; }
    } catch (const std::exception& e) {
        g_model_error = string("parameter ") + name + ": " + e.what();
        return 1;
    }
    return 0;
}

atomdev_t* atom_model_device_handle(atom_model_t* m) { return !m ? nullptr : m->atm ? m->atm->device_handle() : m->hyd->device_handle(); }

double atom_model_mean_temperature_from_curve(atom_model_t* m, double time) {
    if (!m || !m->atm) return NAN;
    try { return m->atm->get_mean_temperature_from_curve((float)time); }
    catch (const std::exception& e) { g_model_error = e.what(); return NAN; }
}

int atom_model_get_tropopause(atom_model_t* m, int* levels, int n) {
    if (!m || !m->atm || !levels) return -1;
    const int count = n < cAtmosphereModel::jm ? (n < 0 ? 0 : n) : cAtmosphereModel::jm;
    const int* t = m->atm->get_tropopause();
    for (int j = 0; j < count; j++) levels[j] = t[j];
    return count;
}

double atom_model_calculate_mean_temperature(atom_model_t* m) {
    if (!m || !m->atm) return NAN;
    try { return m->atm->calculate_mean_temperature(); }
    catch (const std::exception& e) { g_model_error = e.what(); return NAN; }
}

}  // extern "C"
