// Base-class bodies (reference: src/cognitive_engine.cpp:4-6 — all three are empty).
#include "cognitive_engine.hpp"

CognitiveEngine::CognitiveEngine() {}
CognitiveEngine::~CognitiveEngine() {}
void CognitiveEngine::execute() {}
