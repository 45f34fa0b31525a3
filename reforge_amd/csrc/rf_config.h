// rf_config.h -- reforge's pipeline-config DSL, re-stated in C++ (host only).
// Reference: src/config/config_grammar.lalrpop:7-81 (grammar),
//            src/config/config.rs:17-38 (types), :98-205 (parse), :59-90 (type resolution).
#pragma once

#include <map>
#include <string>
#include <vector>

namespace rf {

extern const char* const kFileInput;    // "rf:file-input"   pipeline_graph.rs:22
extern const char* const kFinalOutput;  // "rf:final-output" pipeline_graph.rs:23

// config.rs:17-21
struct ConfigDescriptor {
    std::string resource_name;
    std::string descriptor_name;
};

// config.rs:23-28 (file_path is replaced by the registry lookup of the node type)
struct GraphPipeline {
    std::vector<ConfigDescriptor> inputs;
    std::vector<ConfigDescriptor> outputs;
};

// config.rs:30-33
struct PipelineInstance {
    std::string pipeline_type;
    std::map<std::string, std::string> parameters;   // key -> value in string form
};

// config.rs:35-38.  std::map instead of HashMap: iteration is name-sorted.
struct Config {
    std::map<std::string, GraphPipeline> graph_pipelines;
    std::map<std::string, PipelineInstance> pipeline_instances;

    // add_file_paths, config.rs:59-75: the instance's type, else the node's own name
    const std::string& type_of(const std::string& node) const;
    // the instance's parameters, empty if the node has no instance
    const std::map<std::string, std::string>& params_of(const std::string& node) const;
};

// config::parse, config.rs:98-205.  false + `err` where the reference returns None
// after a warnln!.
bool parse_config(const std::string& text, bool expects_input, Config& out, std::string& err);

// the syntax tree alone, as JSON (rf_config.cpp): what the generated parser returns at config.rs:105
bool parse_syntax(const std::string& text, std::string& json, std::string& err);

// config::single_shader_parse, config.rs:77-90
bool single_node_config(const std::string& type_name, bool expects_input, Config& out, std::string& err);

}  // namespace rf
