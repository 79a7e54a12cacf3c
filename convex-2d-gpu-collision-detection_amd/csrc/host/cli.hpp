// cli.hpp — small command-line parser with the option grammar the reference gets from
// boost::program_options (absent in this image): "--name value", "--name=value", short
// aliases ("-n 5"), value-less switches, and multitoken float lists ("--min_pose 0.1 0.1 0").
#pragma once

#include <cstdlib>
#include <iostream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace cli {

struct Option {
    std::string name;
    char short_name;   // 0 = none
    enum Kind { VALUE, SWITCH, MULTI } kind;
    std::string help;
};

class Parser {
public:
    void add(const std::string& name, char short_name, Option::Kind kind, const std::string& help)
    {
        opts_.push_back(Option{name, short_name, kind, help});
    }
    void parse(int argc, char** argv)
    {
        for (int i = 1; i < argc; i++) {
            std::string tok = argv[i];
            const Option* o = nullptr;
            std::string inline_val;
            bool has_inline = false;
            if (tok.rfind("--", 0) == 0) {
                std::string body = tok.substr(2);
                size_t eq = body.find('=');
                if (eq != std::string::npos) { inline_val = body.substr(eq + 1); body = body.substr(0, eq); has_inline = true; }
                o = find_long(body);
            } else if (tok.size() == 2 && tok[0] == '-' && !is_number(tok)) {
                o = find_short(tok[1]);
            }
            if (!o) throw std::runtime_error("unrecognised option '" + tok + "'");
            std::vector<std::string>& vals = values_[o->name];
            if (o->kind == Option::SWITCH) { vals.push_back("true"); continue; }
            if (has_inline) { vals.push_back(inline_val); if (o->kind == Option::VALUE) continue; }
            if (o->kind == Option::VALUE) {
                if (i + 1 >= argc) throw std::runtime_error("option '" + tok + "' needs a value");
                vals.assign(1, argv[++i]);
            } else {
                while (i + 1 < argc && !looks_like_option(argv[i + 1])) vals.push_back(argv[++i]);
                if (vals.empty()) throw std::runtime_error("option '" + tok + "' needs at least one value");
            }
        }
    }
    bool has(const std::string& name) const { return values_.count(name) != 0; }
    std::string str(const std::string& name) const { return values_.at(name).back(); }
    int integer(const std::string& name) const { return std::stoi(str(name)); }
    long long integer64(const std::string& name) const { return std::stoll(str(name)); }
    float real(const std::string& name) const { return std::stof(str(name)); }
    bool boolean(const std::string& name) const
    {
        const std::string v = str(name);
        if (v == "1" || v == "true" || v == "True" || v == "yes" || v == "on") return true;
        if (v == "0" || v == "false" || v == "False" || v == "no" || v == "off") return false;
        throw std::runtime_error("option '--" + name + "': bad boolean '" + v + "'");
    }
    std::vector<float> reals(const std::string& name) const
    {
        std::vector<float> out;
        for (const auto& v : values_.at(name)) out.push_back(std::stof(v));
        return out;
    }
    void print_help(std::ostream& os) const
    {
        os << "Allowed options:\n";
        for (const auto& o : opts_) {
            std::string left = "  --" + o.name;
            if (o.short_name) left = "  -" + std::string(1, o.short_name) + " [ --" + o.name + " ]";
            if (o.kind != Option::SWITCH) left += " arg";
            if (left.size() < 34) left.append(34 - left.size(), ' ');
            os << left << " " << o.help << "\n";
        }
    }

private:
    static bool is_number(const std::string& s)
    {
        char* end = nullptr;
        std::strtod(s.c_str(), &end);
        return end && *end == '\0' && !s.empty();
    }
    bool looks_like_option(const std::string& s) const
    {
        if (s.rfind("--", 0) == 0) return true;
        return s.size() == 2 && s[0] == '-' && !is_number(s) && find_short(s[1]) != nullptr;
    }
    const Option* find_long(const std::string& n) const
    {
        for (const auto& o : opts_) if (o.name == n) return &o;
        return nullptr;
    }
    const Option* find_short(char c) const
    {
        for (const auto& o : opts_) if (o.short_name == c) return &o;
        return nullptr;
    }
    std::vector<Option> opts_;
    std::map<std::string, std::vector<std::string>> values_;
};

}  // namespace cli
