// main_stub.cpp -- every bin/<tool> is this file compiled with -DASLP_TOOL_MAIN=Main_<tool name with _ for ->, linked to the
// object that implements the tool (model_tools / frame_tools / stream_tools / forward_tools / parallel/worker_tools).
int ASLP_TOOL_MAIN(int argc, char *argv[]);
int main(int argc, char *argv[]) { return ASLP_TOOL_MAIN(argc, argv); }
