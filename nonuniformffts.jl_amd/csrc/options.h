// Development switches of the library (A/B experiments, tests of rarely taken paths; the defaults are the measured optimum).
//
// They reach the library through nufft_params.options — "NUFFT_NAME=value;NUFFT_OTHER=value" — i.e. per plan and visible in the call
// that creates it; the plan keeps its own copy.  The library reads NO environment variable: a process-wide, invisible setting has no
// place in a library that another package loads.  Only a development build (make EXTRA=-DNUFFT_ENV_SWITCHES=1)
// falls back to getenv for names the options string does not hold.
// (The Python development harness, nonuniformffts.jl_amd/plan.py, forwards NUFFT_* environment variables of ITS process as this
// string, so the helper scripts and tests keep their command lines; the Julia binding forwards nothing.)
#pragma once

#include <map>
#include <string>

namespace nufft {

struct Options {
    std::map<std::string, std::string> kv;
    void parse(const char* text);            // "A=1;B=2" (also ',' or whitespace between entries); later entries win
    std::string str() const;                 // canonical form, sorted by name
};

// Options the calling thread currently works under: those of the plan being created, or of the plan the current API call was made on
// (set at the top of every entry point that can reach a switch; cleared when that plan is destroyed).
void set_current_options(const Options* o);
const Options* current_options();

int option_int(const char* name, int fallback);
double option_double(const char* name, double fallback);
bool option_present(const char* name);
const char* option_str(const char* name);    // nullptr when absent

}  // namespace nufft
