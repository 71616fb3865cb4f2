// cognitive_engine.hpp — the CRTS plugin base class, as the reference declares it
// (reference: include/cognitive_engine.hpp:21-45; its three member bodies are empty,
// src/cognitive_engine.cpp:4-6, and live in engine_harness.cpp here).
//
// Engines are constructed by ExtensibleCognitiveRadio::set_ce as
//   new CE_X(int argc, char **argv, ExtensibleCognitiveRadio *ecr)
// (reference: src/extensible_cognitive_radio.cpp:354-369) and driven by the CE worker thread, which
// calls CE->execute() with CE_mutex held (reference: src/extensible_cognitive_radio.cpp:1792-1803).
// Layout contract kept bit-for-bit: one public data member `ECR`, one virtual `execute()`, a
// NON-virtual destructor (the ECR never deletes its engine).  CE_Predictive_Node_GPU.cpp also
// builds against the reference's own copy of this header; tests link it with the reference's own
// CognitiveEngine object code (oracle/_ref/libcrts_ce_base.so).
#ifndef _CE_HPP_
#define _CE_HPP_

class ExtensibleCognitiveRadio;

class CognitiveEngine {
public:
  CognitiveEngine();
  ~CognitiveEngine();
  ExtensibleCognitiveRadio *ECR;
  virtual void execute();
};

#endif
