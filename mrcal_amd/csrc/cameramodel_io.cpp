// The C-level .cameramodel reader and writer: mrcal_read_cameramodel_string/_file,
// the _into variants, mrcal_free_cameramodel, mrcal_write_cameramodel_file
// (reference: mrcal.h:858-890; the reference's reader is a re2c scanner,
// cameramodel-parser.re:356-790, its writer mrcal.c:6626-6680).
//
// A .cameramodel file is a python dict literal: quoted keys (', " or b'..'),
// values that are strings, numbers or (nested) lists / tuples, '#' comments,
// optional trailing commas. Of all keys only lensmodel, intrinsics, imagersize
// and extrinsics / rt_cam_ref are read here; everything else (the
// valid-intrinsics region, the optimization inputs, ...) is skipped over, as
// the reference's C reader does. Host code: this is file I/O.
#include "../../include/mrcal_amd.h"
#include "host_state.hpp"

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <locale.h>

namespace
{
// the "C" locale, whatever LC_NUMERIC the process runs under: numbers in a .cameramodel are Python literals
locale_t c_locale()
{
    static locale_t loc = newlocale(LC_ALL_MASK, "C", (locale_t)0);
    return loc;
}

using mrcal_amd::set_error;

struct Scanner
{
    const char* s;      // 0-terminated
    size_t      i = 0;

    // whitespace and comments
    void skip()
    {
        for(;;)
        {
            const char c = s[i];
            if(c == ' ' || c == '\t' || c == '\n' || c == '\r') i++;
            else if(c == '#') { while(s[i] != '\0' && s[i] != '\n') i++; }
            else return;
        }
    }
    bool take(char c) { skip(); if(s[i] != c) return false; i++; return true; }

    // 'xxx', "xxx", b'xxx', b"xxx". No escapes are interpreted: a backslash
    // protects the next character from ending the string, nothing else
    bool string(std::string* out)
    {
        skip();
        size_t j = i;
        if(s[j] == 'b') j++;
        const char q = s[j];
        if(q != '\'' && q != '"') return false;
        j++;
        const size_t start = j;
        while(s[j] != '\0' && s[j] != q)
        {
            if(s[j] == '\\' && s[j+1] != '\0') j++;
            j++;
        }
        if(s[j] != q) return false;
        if(out) out->assign(s + start, j - start);
        i = j + 1;
        return true;
    }
    // a python float or int literal, as strtod reads it; the token must end there
    bool number(double* out)
    {
        skip();
        const char c = s[i];
        if(!(c == '-' || c == '+' || c == '.' || (c >= '0' && c <= '9'))) return false;
        // decimal digits, a point, an exponent: what Python's literal_eval takes for a number and the reference's
        // scanner (cameramodel-parser.re) accepts. strtod() would also read hex floats, "inf", "nan", and takes its
        // decimal separator from LC_NUMERIC: the token is checked first and converted in the C locale
        size_t j = i;
        while((s[j] >= '0' && s[j] <= '9') || s[j] == '+' || s[j] == '-' || s[j] == '.' || s[j] == 'e' || s[j] == 'E') j++;
        if(s[j] == 'x' || s[j] == 'X' || s[j] == 'p' || s[j] == 'P') return false;
        char* end = NULL;
        const double v = strtod_l(s + i, &end, c_locale());
        if(end == s + i || (size_t)(end - s) > j) return false;
        const char e = *end;
        if(!(e == '\0' || e == ' ' || e == '\t' || e == '\n' || e == '\r' || e == ',' || e == ']' || e == ')' || e == '}' || e == '#'))
            return false;
        if(out) *out = v;
        i = (size_t)(end - s);
        return true;
    }
    // a bare word (None, True, False)
    bool word()
    {
        skip();
        size_t j = i;
        while((s[j] >= 'a' && s[j] <= 'z') || (s[j] >= 'A' && s[j] <= 'Z') || s[j] == '_') j++;
        if(j == i) return false;
        i = j;
        return true;
    }
    // a flat list or tuple of numbers; exactly N of them (N < 0: any number)
    bool list_of_numbers(std::vector<double>* out, int N)
    {
        skip();
        const char open = s[i];
        if(open != '[' && open != '(') return false;
        const char close = (open == '[') ? ']' : ')';
        i++;
        out->clear();
        for(;;)
        {
            if(take(close)) break;
            double v;
            if(!number(&v)) return false;
            out->push_back(v);
            if(take(',')) continue;
            if(take(close)) break;
            return false;
        }
        return N < 0 || (int)out->size() == N;
    }
    // any value: string, number, word, or a balanced (nested) list / tuple / dict
    bool any_value(int depth = 0)
    {
        if(depth > 64) return false;
        skip();
        const char c = s[i];
        if(c == '[' || c == '(' || c == '{')
        {
            const char close = (c == '[') ? ']' : (c == '(') ? ')' : '}';
            i++;
            for(;;)
            {
                if(take(close)) return true;
                if(!any_value(depth + 1)) return false;
                if(c == '{')
                {
                    if(!take(':')) return false;
                    if(!any_value(depth + 1)) return false;
                }
                if(take(',')) continue;
                if(take(close)) return true;
                return false;
            }
        }
        return string(NULL) || number(NULL) || word();
    }
};

constexpr size_t HEADER_BYTES = sizeof(mrcal_cameramodel_VOID_t);

// model != NULL: a caller's buffer with room for *Nintrinsics_max intrinsics
mrcal_cameramodel_VOID_t* parse(mrcal_cameramodel_VOID_t* model, int* Nintrinsics_max, const char* text)
{
    bool reported_size = false;
    mrcal_cameramodel_VOID_t* result = NULL;
    mrcal_cameramodel_VOID_t* allocated = NULL;

    mrcal_lensmodel_t lensmodel; memset(&lensmodel, 0, sizeof(lensmodel));
    lensmodel.type = MRCAL_LENSMODEL_INVALID;
    bool have_lensmodel = false, have_intrinsics = false, have_rt = false, have_size = false;
    std::vector<double> intrinsics, rt, rt2, size;

    Scanner sc{text};
    auto fail = [&](const char* what) { set_error("cameramodel: %s (at byte %zu)", what, sc.i); };

    if(!sc.take('{')) { fail("no leading '{'"); goto done; }
    for(;;)
    {
        // an empty dict, or a trailing comma before the closing brace
        if(sc.take('}')) break;
        std::string key;
        if(!sc.string(&key)) { fail("expected a quoted key"); goto done; }
        if(!sc.take(':'))    { fail("expected ':' after a key"); goto done; }

        if(key == "lensmodel")
        {
            if(have_lensmodel) { fail("lensmodel defined more than once"); goto done; }
            std::string name;
            if(!sc.string(&name)) { fail("lensmodel must be a string"); goto done; }
            if(!mrcal_lensmodel_from_name(&lensmodel, name.c_str()))
            { set_error("cameramodel: could not parse lensmodel '%s'", name.c_str()); goto done; }
            have_lensmodel = true;
        }
        else if(key == "intrinsics")
        {
            if(have_intrinsics) { fail("intrinsics defined more than once"); goto done; }
            // (the reference insists on the order too: the count comes from the model)
            if(!have_lensmodel) { fail("'intrinsics' before 'lensmodel': the lensmodel key must come first"); goto done; }
            const int N = mrcal_lensmodel_num_params(&lensmodel);
            if(model != NULL && N > *Nintrinsics_max)
            {
                *Nintrinsics_max = N;
                reported_size = true;
                // not an error to shout about: the caller asked how much room is needed
                mrcal_amd::last_error_string() = "cameramodel: the buffer is too small for this model's intrinsics";
                goto done;
            }
            if(!sc.list_of_numbers(&intrinsics, N)) { fail("intrinsics: expected a list of exactly as many numbers as the lens model has parameters"); goto done; }
            have_intrinsics = true;
        }
        else if(key == "extrinsics" || key == "rt_cam_ref")
        {
            if(!have_rt)
            {
                if(!sc.list_of_numbers(&rt, 6)) { fail("extrinsics: expected a list of 6 numbers"); goto done; }
                have_rt = true;
            }
            else
            {
                // both the old and the new name may be there; they must agree
                if(!sc.list_of_numbers(&rt2, 6)) { fail("extrinsics: expected a list of 6 numbers"); goto done; }
                for(int k = 0; k < 6; k++)
                    if(fabs(rt[k] - rt2[k]) > 1e-9) { fail("extrinsics defined more than once, differently"); goto done; }
            }
        }
        else if(key == "imagersize")
        {
            if(have_size) { fail("imagersize defined more than once"); goto done; }
            if(!sc.list_of_numbers(&size, 2) ||
               !(size[0] > 0 && size[1] > 0 && size[0] == floor(size[0]) && size[1] == floor(size[1]) && size[0] < 4e9 && size[1] < 4e9))
            { fail("imagersize: expected two positive integers"); goto done; }
            have_size = true;
        }
        else if(!sc.any_value()) { fail("could not read the value of an unknown key"); goto done; }

        if(sc.take(',')) continue;
        if(sc.take('}')) break;
        fail("expected ',' or '}' after a value");
        goto done;
    }
    sc.skip();
    if(sc.s[sc.i] != '\0') { fail("garbage after the closing '}'"); goto done; }
    if(!(have_lensmodel && have_intrinsics && have_rt && have_size))
    { set_error("cameramodel: lensmodel, intrinsics, extrinsics (rt_cam_ref) and imagersize are all required"); goto done; }

    if(model == NULL)
    {
        allocated = (mrcal_cameramodel_VOID_t*)malloc(HEADER_BYTES + intrinsics.size()*sizeof(double));
        if(allocated == NULL) { set_error("cameramodel: malloc() failed"); goto done; }
        model = allocated;
    }
    memset(model, 0, HEADER_BYTES);
    for(int k = 0; k < 6; k++) model->rt_cam_ref[k] = rt[k];
    model->imagersize[0] = (unsigned int)size[0];
    model->imagersize[1] = (unsigned int)size[1];
    model->lensmodel     = lensmodel;
    memcpy(model->intrinsics, intrinsics.data(), intrinsics.size()*sizeof(double));
    result = model;

 done:
    if(Nintrinsics_max != NULL && !reported_size && result == NULL) *Nintrinsics_max = 0;
    return result;
}

// len > 0: not necessarily terminated
mrcal_cameramodel_VOID_t* parse_buffer(mrcal_cameramodel_VOID_t* model, int* Nintrinsics_max, const char* string, int len)
{
    if(string == NULL) { set_error("cameramodel: NULL string"); if(Nintrinsics_max) *Nintrinsics_max = 0; return NULL; }
    if(len <= 0) return parse(model, Nintrinsics_max, string);
    // an embedded 0 byte would end the text early: take everything in front of it
    const std::string copy(string, strnlen(string, (size_t)len));
    return parse(model, Nintrinsics_max, copy.c_str());
}

bool slurp(std::string* out, const char* filename)
{
    FILE* fp = (filename != NULL) ? fopen(filename, "rb") : NULL;
    if(fp == NULL) { set_error("cameramodel: could not open '%s'", filename ? filename : "(null)"); return false; }
    char buf[65536];
    size_t n;
    out->clear();
    while((n = fread(buf, 1, sizeof(buf), fp)) > 0) out->append(buf, n);
    const bool ok = !ferror(fp);
    fclose(fp);
    if(!ok) set_error("cameramodel: error reading '%s'", filename);
    return ok;
}
}

extern "C"
{
mrcal_cameramodel_VOID_t* mrcal_read_cameramodel_string(const char* string, const int len)
{
    mrcal_cameramodel_VOID_t* m = parse_buffer(NULL, NULL, string, len);
    return m;
}
mrcal_cameramodel_VOID_t* mrcal_read_cameramodel_file(const char* filename)
{
    std::string text;
    mrcal_cameramodel_VOID_t* m = slurp(&text, filename) ? parse(NULL, NULL, text.c_str()) : NULL;
    return m;
}
void mrcal_free_cameramodel(mrcal_cameramodel_VOID_t** cameramodel)
{
    if(cameramodel == NULL) return;
    free(*cameramodel);
    *cameramodel = NULL;
}
bool mrcal_read_cameramodel_string_into(mrcal_cameramodel_VOID_t* model, int* Nintrinsics_max, const char* string, const int len)
{
    if(model == NULL || Nintrinsics_max == NULL) { set_error("cameramodel: NULL output"); return false; }
    const bool ok = parse_buffer(model, Nintrinsics_max, string, len) != NULL;
    return ok;
}
bool mrcal_read_cameramodel_file_into(mrcal_cameramodel_VOID_t* model, int* Nintrinsics_max, const char* filename)
{
    if(model == NULL || Nintrinsics_max == NULL) { set_error("cameramodel: NULL output"); return false; }
    std::string text;
    if(!slurp(&text, filename)) { *Nintrinsics_max = 0; return false; }
    const bool ok = parse(model, Nintrinsics_max, text.c_str()) != NULL;
    return ok;
}

// Full precision (%.17g round-trips a double; the reference prints %f, which
// loses small distortion coefficients), both names of the extrinsics, as the
// python writer emits them
bool mrcal_write_cameramodel_file(const char* filename, const mrcal_cameramodel_VOID_t* cameramodel)
{
    if(filename == NULL || cameramodel == NULL) { set_error("cameramodel: NULL argument"); return false; }
    char name[1024];
    if(!mrcal_lensmodel_name(name, sizeof(name), &cameramodel->lensmodel))
    { set_error("cameramodel: cannot name the lens model (type %d)", (int)cameramodel->lensmodel.type); return false; }
    const int N = mrcal_lensmodel_num_params(&cameramodel->lensmodel);
    if(N < 0) { set_error("cameramodel: lens model '%s' has no parameter count", name); return false; }
    FILE* fp = fopen(filename, "w");
    if(fp == NULL) { set_error("cameramodel: could not open '%s' for writing", filename); return false; }
    // (printf takes its decimal separator from LC_NUMERIC: a comma there would split every number in two for
    //  Python's literal_eval and for the reader above)
    const locale_t previous = uselocale(c_locale());
    fprintf(fp, "{\n    'lensmodel':  '%s',\n\n    'intrinsics': [", name);
    for(int i = 0; i < N; i++) fprintf(fp, " %.17g,", cameramodel->intrinsics[i]);
    fprintf(fp, "],\n\n");
    for(const char* key : {"rt_cam_ref", "extrinsics"})
    {
        fprintf(fp, "    '%s': [", key);
        for(int i = 0; i < 6; i++) fprintf(fp, " %.17g,", cameramodel->rt_cam_ref[i]);
        fprintf(fp, "],\n");
    }
    fprintf(fp, "\n    'imagersize': [ %u, %u ],\n}\n", cameramodel->imagersize[0], cameramodel->imagersize[1]);
    uselocale(previous);
    const bool ok = !ferror(fp);
    if(fclose(fp) != 0 || !ok) { set_error("cameramodel: error writing '%s'", filename); return false; }
    return true;
}
}
