/*
 * atom_models_c.h - the models' outer interface (include/atom_models.hpp) behind plain C calls, for bindings from other
 * languages: what the reference exposes to Python through python/atom.pxd and python/pyatom.pyx.template:44-188 (create,
 * load_config, run, run_time_slice, one property per configuration parameter).  Parameters travel as text: "true" /
 * "false", decimal integers, doubles with 17 significant digits, strings as they are.  Calls return 0 or an error code;
 * atom_model_last_error() has the message (errors of the C++ classes are exceptions, caught here).
 */
#ifndef ATOM_MODELS_C_H_
#define ATOM_MODELS_C_H_

#include <stddef.h>

#include "atomdev.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct atom_model atom_model_t;

const char* atom_model_last_error(void);
atom_model_t* atom_model_create(int model);       /* ATOMDEV_ATM: cAtmosphereModel, ATOMDEV_HYD: cHydrosphereModel */
void atom_model_destroy(atom_model_t* m);
int atom_model_load_config(atom_model_t* m, const char* filename);   /* LoadConfig */
int atom_model_run(atom_model_t* m);                                  /* Run */
int atom_model_run_time_slice(atom_model_t* m, int Ma);               /* RunTimeSlice */

/* The configuration parameters of a model in the order of the reference's generated headers: <common>, then its own. */
int atom_model_param_count(int model);
const char* atom_model_param_name(int model, int index);
const char* atom_model_param_type(int model, int index);      /* "int", "double", "bool", "string" */
const char* atom_model_param_section(int model, int index);   /* "common", "atmosphere", "hydrosphere" */
int atom_model_get_param(atom_model_t* m, const char* name, char* buffer, size_t size);
int atom_model_set_param(atom_model_t* m, const char* name, const char* text);

/* The device handle that holds the model's fields: NULL before the first time slice. */
atomdev_t* atom_model_device_handle(atom_model_t* m);
/* cAtmosphereModel::get_mean_temperature_from_curve (atmosphere handles; NaN otherwise or outside the curve) */
double atom_model_mean_temperature_from_curve(atom_model_t* m, double time);
/* cAtmosphereModel::get_tropopause (cAtmosphereModel.h:80-82): copies the jm = 181 tropopause levels to levels[0 .. n-1],
 * n <= 181; returns the number copied, -1 for a model that is no atmosphere. */
int atom_model_get_tropopause(atom_model_t* m, int* levels, int n);
/* cAtmosphereModel::calculate_mean_temperature (cAtmosphereModel.h:84, 108-110) of the model's own temperature field: the
 * cosine-weighted mean of the sea-level temperature in C (NaN before the first time slice or for a hydrosphere). */
double atom_model_calculate_mean_temperature(atom_model_t* m);

#ifdef __cplusplus
}
#endif
#endif /* ATOM_MODELS_C_H_ */
